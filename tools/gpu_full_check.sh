#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the whole GPU test-suite, then the default bench line.
cd "$(dirname "$0")/.." && REPO=$PWD
OUT=$REPO/gpurun_out/full_check; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-6000
tail -5 $OUT/bench_default.err
