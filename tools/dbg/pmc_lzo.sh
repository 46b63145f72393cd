#!/bin/bash
# instruction mix of the LZO kernels on C5-like batches (run on the GPU box): bash tools/dbg/pmc_lzo.sh [bench_lzo args]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_lzo
rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/tools/bench_lzo.py ${*:---streams 4096 --kind text}"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/$tag -o pmc -- $BENCH > /dev/null 2> $OUT/$tag.log
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "md::lzo" not in k: continue
        k = k.split("(")[0].split("::")[-1]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-26s %.4g" % (c, sum(v) / len(v)))
PY
rm -rf $OUT/SQ_*
