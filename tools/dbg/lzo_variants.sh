#!/bin/bash
# round 6: the LZO compressor's build-time knobs side by side (tools/dbg/variants/*.so): parity tests + timings each
mkdir -p gpurun_out/lzo_iter
for so in tools/dbg/variants/*.so; do
  MD_LIBMDEFLATE=$PWD/$so timeout 600 python -m pytest tests/test_gpu_lzo.py -x -q 2>&1 | tail -1
  for a in "--streams 8192" "--streams 4096 --kind text" "--streams 256 --kind text"; do
    MD_LIBMDEFLATE=$PWD/$so timeout 300 python tools/bench_lzo.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$so $a:', d['compress_ms'], d['uncompress_ms'], d['parity_ok'])"
  done
done 2>&1 | tee gpurun_out/lzo_iter/variants.txt
