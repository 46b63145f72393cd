#!/bin/bash
# instruction mix / stall / HBM counters of the three deflate kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_deflate
mkdir -p $OUT
BENCH="python $REPO/tools/bench_deflate.py --streams ${STREAMS:-2048} --stream-kib ${KIB:-1024} --level 6 --steps 1 $*"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
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
