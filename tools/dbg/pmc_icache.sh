#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_icache
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -oE "(SQC_ICACHE|SQ_IFETCH|SQ_LEVEL_WAVES|SQ_INST_LEVEL|SQ_WAVES|SQC_DCACHE|SQ_INSTS_SMEM|SQ_ACCUM_PREV|SQ_BUSY_CU_CYCLES|SQ_VALU_MFMA_BUSY|SQ_ACTIVE_INST_MISC|SQ_INST_CYCLES_SALU|SQ_THREAD_CYCLES_VALU|SQ_WAIT_ANY)[A-Z_0-9]*" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt | tr '\n' ' '
BENCH="python $REPO/tools/bench_deflate.py --streams ${STREAMS:-4096} --stream-kib ${KIB:-256} --level 6 --steps 1"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQ_IFETCH" "SQ_LEVEL_WAVES SQ_WAVES" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/$tag -o pmc -- $BENCH > /dev/null 2> $OUT/$tag.log
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "md::" not in k: continue
        k = k.split("(")[0].split("<")[0].split("::")[-1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for k, d in agg.items():
        print(k); out.write(k + "\n")
        for c, v in sorted(d.items()):
            line = "   %-26s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v))
            print(line); out.write(line + "\n")
PY
