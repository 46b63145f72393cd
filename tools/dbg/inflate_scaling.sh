#!/bin/bash
# Run ON THE GPU BOX: inflate kernel time against the number of streams (how many fit the chip at once, what a second generation costs)
cd "$(dirname "$0")/../.."
for n in 1024 2048 3072 4096 8192; do
  for mode in pair single; do
    W=2; [ $mode = single ] && W=1
    timeout 300 python bench.py --inflate-waves $W --streams $n --steps 5 --warmup 1 --no-cpu-baseline --no-text-leg --no-secondary --no-deflate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n $mode', d['ms_per_step'], d['parity_ok'])"
  done
done
