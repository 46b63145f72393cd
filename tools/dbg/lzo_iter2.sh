#!/bin/bash
mkdir -p gpurun_out/lzo_iter
for a in "--streams 8192" "--streams 8192 --kind blocks" "--streams 8192 --kind text" "--streams 2048 --kind text"; do
  timeout 300 python tools/bench_lzo.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a:', d['compress_ms'], d['uncompress_ms'], d['parity_ok'], d['ratio'])"
done 2>&1 | tee gpurun_out/lzo_iter/bench2.txt
