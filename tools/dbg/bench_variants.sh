#!/bin/bash
# Run ON THE GPU BOX: the headline leg for every decompress_amd/libmdeflate_v*.so next to the product library
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/variants; mkdir -p $OUT
for so in decompress_amd/libmdeflate.so decompress_amd/libmdeflate_v*.so; do
  name=$(basename $so .so)
  MD_LIBMDEFLATE=$REPO/$so timeout 600 python bench.py --no-deflate --no-secondary --no-cpu-baseline --no-host-path ${BENCH_ARGS} > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/variants/%s.json'%n).readline())
    print(n, 'C2 kernel', d['roofline']['kernel_ms'], 'parity', d['parity_ok'], '| r01', d.get('r01_workload',{}).get('kernel_ms'))
except Exception as e:
    print(n, 'FAILED', e)
PY
done
