#!/bin/bash
# Run ON THE GPU BOX: instruction mix / wait counters of the deflate kernels on word text (level 6, 4096 x 256 KiB: the chip is full)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_text; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/tools/bench_deflate.py --streams 4096 --stream-kib 256 --level ${LEVEL:-6} --kind text --steps 2"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/$tag -o pmc -- $BENCH > /dev/null 2> $OUT/$tag.log
done
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/t -o trace -- $BENCH 2> /dev/null | tail -1 | cut -c1-200
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "md::" not in k: continue
        k = k.split("<")[0].split("(")[0].split("::")[-1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_stats*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'md::' in r['Name']: print('   ', r['Name'].split('(')[0][-40:], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
