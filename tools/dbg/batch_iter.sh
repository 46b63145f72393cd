#!/bin/bash
cd "$(dirname "$0")/../.." && REPO=$PWD
OUT=$REPO/gpurun_out/batch_iter; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_deflate.py -x -q -m gpu -k "many_encoders or encoder_in_pieces or streaming_encoder" 2>&1 | tail -25 > $OUT/pytest.txt
cat $OUT/pytest.txt
