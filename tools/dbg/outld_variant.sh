#!/bin/bash
# round 6: output re-reads (match sources) as agent-scope loads instead of non-temporal: C2 inflate + LZO, both builds
mkdir -p gpurun_out/outld
for so in decompress_amd/libmdeflate.so tools/dbg/variants/lib_outld1.so; do
  echo "== $so"
  MD_LIBMDEFLATE=$PWD/$so timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-text-leg --no-secondary --no-host-path --no-deflate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C2 inflate ms', d['ms_per_step'], 'parity', d['parity_ok'])"
  MD_LIBMDEFLATE=$PWD/$so timeout 300 python tools/bench_lzo.py --streams 8192 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lzo', d['compress_ms'], d['uncompress_ms'], d['parity_ok'])"
done 2>&1 | tee gpurun_out/outld/result.txt
