#!/bin/bash
# Run ON THE GPU BOX: per-kernel times of the De.Def.Ns path
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/def_ns_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o trace -- python $REPO/tools/bench_def_ns.py 2> $OUT/err.txt | tail -1 | cut -c1-420
find $OUT/t -name '*kernel_stats*.csv' -exec cp {} $OUT/stats.csv \;
python - $OUT/stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'md::' in r['Name']: print('   ', r['Name'].split('(')[0][-44:], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
rm -rf $OUT/t
