#!/bin/bash
# Run ON THE GPU BOX: inflate kernel time against the number of streams, from one stream per CU up
cd "$(dirname "$0")/../.."
for n in 256 512 1024 2048 4096; do
  timeout 300 python bench.py --streams $n --steps 5 --warmup 1 --no-cpu-baseline --no-text-leg --no-secondary --no-deflate --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n streams:', d['roofline']['kernel_ms'], 'ms', d['parity_ok'])"
done
