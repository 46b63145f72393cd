#!/bin/bash
# Run ON THE GPU BOX (via gpurun): deflate parity tests, then timings + a kernel trace of the three deflate kernels.
cd "$(dirname "$0")/.." && REPO=$PWD
OUT=$REPO/gpurun_out/deflate_check; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_def_ns.py tests/test_gpu_deflate.py tests/test_gpu_gzip.py tests/test_cli.py tests/test_c_consumer.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
for args in "--streams 2048 --stream-kib 1024 --level 6" "--streams 4096 --stream-kib 256 --level 6" "--streams 1024 --stream-kib 256 --level 6 --kind text" "--streams 1024 --stream-kib 256 --level 4 --kind text"; do
  timeout 600 python tools/bench_deflate.py $args 2>&1 | tail -1 | tee -a $OUT/bench.jsonl
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-text-leg --no-secondary > $OUT/bench_c2c3.json 2> $OUT/trace_run.txt
tail -1 $OUT/bench_c2c3.json | cut -c1-1500
find $OUT/trace -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats.csv \;
python - <<'PY'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/deflate_check/kernel_stats.csv')):
    if 'md::' in r['Name']: print(r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
cd $REPO && timeout 300 python tools/dbg/deflate_prof.py 2048 1024 ascii 2>&1 | tail -20
