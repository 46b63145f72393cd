#!/bin/bash
# round 6: LZO parity tests + the C5 timing, one GPU call
mkdir -p gpurun_out/lzo_iter
timeout 600 python -m pytest tests/test_gpu_lzo.py -x -q > gpurun_out/lzo_iter/pytest.txt 2>&1
tail -3 gpurun_out/lzo_iter/pytest.txt
for a in "--streams 8192" "--streams 4096 --kind text" "--streams 256 --kind text"; do
  timeout 300 python tools/bench_lzo.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a:', d['compress_ms'], d['uncompress_ms'], d['parity_ok'], d['ratio'])"
done 2>&1 | tee gpurun_out/lzo_iter/bench.txt
