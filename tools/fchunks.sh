#!/bin/bash
# usage: tools/fchunks.sh "<fwd tile>" "<chunks list>"; forward run-length sweep of the streaming kernel
cd /root/repo
for c in $2; do
  echo -n "chunks=$c "
  VW_FWD_CHUNKS=$c tools/tune.sh "$1:4,1" || exit 1
done
