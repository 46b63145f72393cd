#!/bin/bash
# Run ON THE GPU BOX: variants of libmdeflate.so (gpurun_variants/<name>.so) on C3, C4 and word text at levels 4 / 6
cd "$(dirname "$0")/../.."
for v in "$@"; do
  cp gpurun_variants/$v.so decompress_amd/libmdeflate.so
  echo "== $v"
  timeout 200 python tools/dbg/deflate_slices.py 0 2>&1 | tail -1
  timeout 300 python tools/bench_deflate.py --streams 1024 --stream-kib 256 --level 6 --steps 2 --kind text 2>&1 | tail -1 | cut -c1-200
  timeout 300 python tools/bench_deflate.py --streams 1024 --stream-kib 256 --level 4 --steps 2 --kind text 2>&1 | tail -1 | cut -c1-200
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-leg --no-deflate --no-verify --deflate-cap-mib 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C4 deflate ms', d['gzip']['deflate']['ms'], 'def_ns', d['def_ns']['ms'])
"
done
cp gpurun_variants/base.so decompress_amd/libmdeflate.so
