#!/bin/bash
# Run ON THE GPU BOX: Lzo.compress / uncompress against the number of streams (is a stream's own chain the bound?)
cd "$(dirname "$0")/../.."
for n in 2048 8192 16384 32768; do
  timeout 600 python tools/bench_lzo.py --streams $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n streams:', json.dumps({k:d[k] for k in d if k in ('compress','uncompress','value','parity_ok','ms_compress','ms_uncompress','compress_ms','uncompress_ms')})[:300])"
done
