#!/bin/bash
# Run ON THE GPU BOX: the known-bounds floor again (with the no-wait mode), then the whole GPU suite
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/r05_wall2; mkdir -p $OUT
MD_LIBMDEFLATE=$REPO/decompress_amd/libmdeflate_kb.so timeout 900 python tools/dbg/inflate_floor.py $OUT/inflate_floor.json > $OUT/floor.txt 2>&1
grep "'waves': 2, 'streams': 4096\|'waves': 2, 'streams': 256" $OUT/floor.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
