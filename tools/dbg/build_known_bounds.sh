#!/bin/bash
# Build the measurement variant of the library (inflate kernel with -DMD_DEBUG_KNOWN_BOUNDS: record / replay of the sync
# passes' results, copier ablations) next to the product one.  Use it with MD_LIBMDEFLATE=decompress_amd/libmdeflate_kb.so
cd "$(dirname "$0")/../.."
MD_SO_OUT=$PWD/decompress_amd/libmdeflate_kb.so MD_HIPCC_FLAGS="-DMD_DEBUG_KNOWN_BOUNDS" python -m decompress_amd.build --force
