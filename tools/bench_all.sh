#!/bin/bash
# All benchmark lines of this repo on one MI355X (run on the GPU box).  The headline is bench.py;
# the others are the BASELINE configs 3, 4 (one GPU's share) and 5, and deflate on match-rich text.
cd "$(dirname "$0")/.."
python bench.py 2>/dev/null | tail -1
python tools/bench_deflate.py --streams 4096 --stream-kib 1024 2>/dev/null | tail -1
python tools/bench_deflate.py --streams 1024 --stream-kib 256 --kind text --level 6 2>/dev/null | tail -1
python tools/bench_deflate.py --streams 1024 --stream-kib 256 --kind text --level 4 2>/dev/null | tail -1
python tools/bench_gzip.py --streams 4096 2>/dev/null | tail -1
python tools/bench_lzo.py --streams 8192 2>/dev/null | tail -1
