#!/bin/bash
# instruction mix / stall counters of the inflate pair (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_insts
mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-host-path --unique 256 $*"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/$tag -o pmc -- $BENCH > /dev/null 2> $OUT/$tag.log
done
python - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "md::" not in k: continue
        k = k.split("<")[0].split("::")[-1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %.4g" % (c, sum(v) / len(v)))
PY
