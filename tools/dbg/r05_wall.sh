#!/bin/bash
# Run ON THE GPU BOX: where the wall of the inflate kernel is (occupancy curve, known-bounds floor), then the default bench
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/r05_wall; mkdir -p $OUT
ls oracle/*.so oracle/_ref/*.so decompress_amd/*.so > $OUT/libs.txt 2>&1
timeout 900 python tools/dbg/inflate_occupancy.py $OUT/inflate_occupancy.json > $OUT/occupancy.txt 2>&1
tail -40 $OUT/occupancy.txt
MD_LIBMDEFLATE=$REPO/decompress_amd/libmdeflate_kb.so timeout 900 python tools/dbg/inflate_floor.py $OUT/inflate_floor.json > $OUT/floor.txt 2>&1
tail -50 $OUT/floor.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-3000
tail -3 $OUT/bench_default.err
