#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + HBM PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
# Summaries land in gpurun_out/prof_<tag>/ ; copy what you want judged to profiles/.
set -u
TAG=${1:-r01}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --unique 512 $*"
echo "== kernel trace: $BENCH"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
tail -1 "$OUT/bench_trace.json"
find "$OUT/trace" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# PMC passes (separate runs, no tracing domains besides kernel-trace): HBM bytes
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --kernel-trace --pmc $C -f csv -d "$OUT/pmc_$C" -o pmc -- $BENCH > /dev/null 2> "$OUT/pmc_$C.log"
  find "$OUT/pmc_$C" -name '*counter_collection*.csv' -exec cp {} "$OUT/pmc_$C.csv" \;
done
python - "$OUT" <<'PY'
import csv, sys, os, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, "pmc_%s.csv" % c)
    if not os.path.exists(p):
        print(c, "missing"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r.get("Counter_Name") == c:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%s  %s  launches=%d  mean=%.1f (KiB units per rocprofv3)" % (c, k, len(v), sum(v) / len(v)))
PY
ls -la "$OUT"
