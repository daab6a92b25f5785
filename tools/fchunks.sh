#!/bin/bash
# needs the tuning build (make -C vectorwave_amd/csrc tuning): only that build reads the VW_* overrides
export VW_LIBVWGPU=${VW_LIBVWGPU:-/root/repo/vectorwave_amd/libvwgpu_tuning.so}
# usage: tools/fchunks.sh "<fwd tile>" "<chunks list>"; forward run-length sweep of the streaming kernel
cd /root/repo
for c in $2; do
  echo -n "chunks=$c "
  VW_FWD_CHUNKS=$c tools/tune.sh "$1:4,1" || exit 1
done
