#!/bin/bash
# round 6: the inflate tests + the C2 line, one GPU call
mkdir -p gpurun_out/inflate_quick
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_lzo.py -x -q 2>&1 | tail -3 | tee gpurun_out/inflate_quick/pytest.txt
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-text-leg --no-secondary --no-host-path --no-deflate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C2 inflate ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'parity', d['parity_ok'])"
done | tee gpurun_out/inflate_quick/bench.txt
