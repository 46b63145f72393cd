#!/bin/bash
# Run ON THE GPU BOX: per-kernel times of the deflate path on compressible input (word text, the corpus) at levels 4 and 6
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/front_text; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for args in "--streams 1024 --stream-kib 256 --level 4 --kind text" "--streams 1024 --stream-kib 256 --level 6 --kind text" "--streams 2048 --level 4 --kind corpus" "--streams 2048 --level 6 --kind corpus" "--streams 1024 --stream-kib 1024 --level 6 --kind ascii"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t$i -o trace -- python $REPO/tools/bench_deflate.py $args --steps 3 2> $OUT/err$i.txt | tail -1 | cut -c1-420
  find $OUT/t$i -name '*kernel_stats*.csv' -exec cp {} $OUT/stats$i.csv \;
  python - $OUT/stats$i.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'md::' in r['Name']: print('   ', r['Name'].split('(')[0][-40:], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
  rm -rf $OUT/t$i
done
