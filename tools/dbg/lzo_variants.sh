#!/bin/bash
# round 6: the LZO compressor's build-time knobs side by side (tools/dbg/variants/lib_w<waves>_f<probes>.so)
mkdir -p gpurun_out/lzo_iter
for so in tools/dbg/variants/*.so; do
  for a in "--streams 8192" "--streams 256 --kind text"; do
    MD_LIBMDEFLATE=$PWD/$so timeout 300 python tools/bench_lzo.py $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$so $a:', d['compress_ms'], d['uncompress_ms'], d['parity_ok'])"
  done
done 2>&1 | tee gpurun_out/lzo_iter/variants.txt
