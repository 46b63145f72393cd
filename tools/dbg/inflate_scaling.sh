#!/bin/bash
# how does the inflate pair scale with the number of resident streams? (latency- vs throughput-bound)
for n in 512 1024 2048 3072 3584 4096 6144 8192; do
  python bench.py --streams $n --unique 256 --steps 5 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('%5d streams  %.3f ms  %.1f GiB/s' % (l['config']['streams_per_gpu'], l['roofline']['kernel_ms'], l['value'] / 1024))"
done
