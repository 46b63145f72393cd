#!/bin/bash
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/r05_wall3; mkdir -p $OUT
MD_LIBMDEFLATE=$REPO/decompress_amd/libmdeflate_kb.so timeout 900 python tools/dbg/inflate_floor.py $OUT/inflate_floor.json > $OUT/floor.txt 2>&1
grep "'waves': 2, 'streams': 4096\|'waves': 2, 'streams': 256" $OUT/floor.txt
bash tools/dbg/inflate_iter.sh
