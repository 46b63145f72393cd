#!/bin/bash
cd "$(dirname "$0")/../.."
for cap in "--deflate-cap-mib 40800" "--deflate-cap-mib 36000"; do
  echo "== cap '$cap'"
  timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-leg --no-deflate --no-verify $cap 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
def find(o):
    if isinstance(o, dict):
        if 'deflate' in o and isinstance(o['deflate'], dict) and 'ms' in o['deflate']: print('C4 deflate ms', o['deflate']['ms'], 'inflate', o.get('inflate', {}).get('ms'), o.get('workload', '')[:150])
        for v in o.values(): find(v)
    elif isinstance(o, list):
        for v in o: find(v)
find(d)
"
done
