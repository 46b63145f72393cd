#!/bin/bash
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/host_iter; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu -k "host_entry or c_consumer or golden" 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 900 python bench.py --no-secondary --no-cpu-baseline --no-text-leg > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/host_iter/bench.json').readline())
print('C2', d['ms_per_step'], d['parity_ok'], json.dumps(d.get('host_path')))
print('C3', d['deflate']['ms_per_step'], d['deflate']['parity_ok'], d['deflate']['parity'], json.dumps(d['deflate'].get('host_path')))
PY
tail -5 $OUT/bench.err
