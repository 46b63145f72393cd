#!/bin/bash
cd "$(dirname "$0")/../.."
for v in "$@"; do
  cp gpurun_variants/$v.so decompress_amd/libmdeflate.so
  echo "== $v"
  timeout 200 python tools/dbg/deflate_slices.py 0 2>&1 | tail -1
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-leg --no-deflate --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = [l for l in d.get('secondary', d.get('legs', [])) if 'C4' in str(l)] if isinstance(d.get('secondary', d.get('legs', None)), list) else None
def find(o):
    if isinstance(o, dict):
        if 'deflate' in o and isinstance(o['deflate'], dict) and 'ms' in o['deflate']: print('C4 deflate ms', o['deflate']['ms'], 'inflate', o.get('inflate', {}).get('ms'))
        for v in o.values(): find(v)
    elif isinstance(o, list):
        for v in o: find(v)
find(d)
"
done
cp gpurun_variants/base.so decompress_amd/libmdeflate.so
