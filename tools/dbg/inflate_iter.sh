#!/bin/bash
# Run ON THE GPU BOX: the inflate parity tests, then the headline leg alone (what one iteration on the inflate kernel needs)
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/inflate_iter; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fuzz.py tests/test_gpu_gzip.py -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 600 python bench.py --no-deflate --no-secondary --no-cpu-baseline --profile > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/inflate_iter/bench.json').readline())
print('C2', d['ms_per_step'], 'ms kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'parity', d['parity_ok'], '| r01 workload', d.get('r01_workload',{}).get('ms_per_step'))
PY
tail -30 $OUT/bench.err
