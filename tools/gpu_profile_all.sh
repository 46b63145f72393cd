#!/bin/bash
# Run ON THE GPU BOX: the round's profile set (kernel trace + HBM counters of bench.py, instruction mix of the deflate kernels,
# kernel trace of the gzip / LZO legs)
cd "$(dirname "$0")/.." && REPO=$PWD
TAG=${TAG:-r06_final}
bash tools/profile_gpu.sh $TAG
bash tools/dbg/pmc_deflate.sh > /dev/null 2>&1
cp gpurun_out/pmc_deflate/summary.txt gpurun_out/prof_$TAG/pmc_insts_deflate.txt
bash tools/dbg/pmc_insts.sh --no-deflate --no-text-leg 2> /dev/null | grep -A14 "^inflate_wave_kernel" > gpurun_out/prof_$TAG/pmc_insts_inflate.txt
timeout 1500 python bench.py > gpurun_out/prof_$TAG/bench_full.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof_$TAG/trace2 -o trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-text-leg --no-host-path --no-deflate > $REPO/gpurun_out/prof_$TAG/bench_secondary.json 2> /dev/null
find $REPO/gpurun_out/prof_$TAG/trace2 -name '*kernel_stats*.csv' -exec cp {} $REPO/gpurun_out/prof_$TAG/kernel_stats_secondary.csv \;
rm -rf $REPO/gpurun_out/prof_$TAG/trace2 $REPO/gpurun_out/prof_$TAG/trace $REPO/gpurun_out/prof_$TAG/pmc_FETCH_SIZE $REPO/gpurun_out/prof_$TAG/pmc_WRITE_SIZE
ls -la $REPO/gpurun_out/prof_$TAG
