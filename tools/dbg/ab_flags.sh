#!/bin/bash
# Run ON THE GPU BOX: builds in gpurun_variants/<name>.so on the inflate bench (C2 kernel ms), C3 and text level 4
cd "$(dirname "$0")/../.."
for v in "$@"; do
  cp gpurun_variants/$v.so decompress_amd/libmdeflate.so
  a=$(timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-deflate --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['r01_workload']['kernel_ms'], d['parity_ok'])")
  b=$(timeout 200 python tools/dbg/deflate_slices.py 0 2>&1 | tail -1 | cut -c1-30)
  c=$(timeout 200 python tools/bench_deflate.py --streams 1024 --stream-kib 256 --level 4 --steps 2 --kind text 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['kernel_ms'])")
  echo "$v: inflate C2/r01 $a | C3 $b | text L4 $c"
done
cp gpurun_variants/base.so decompress_amd/libmdeflate.so
