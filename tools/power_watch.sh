#!/bin/bash
# board power and shader clock while a kernel variant runs back to back: tools/power_watch.sh <seconds> <ab_plan variant...>
cd /root/repo
secs=$1; shift
timeout -k 5 $((secs+60)) python tools/ab_plan.py --rounds $((secs*2)) --reps 10 "$@" > /tmp/abw.log 2>&1 &
pid=$!
sleep 8
for i in $(seq 1 8); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
  sleep 1
done
kill $pid 2>/dev/null; wait $pid 2>/dev/null
tail -2 /tmp/abw.log
