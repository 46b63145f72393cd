#!/bin/bash
# build variants of the library with extra -D flags: tools/dbg/build_variants.sh name1 "flags1" name2 "flags2" ...
cd "$(dirname "$0")/../.."
while [ $# -ge 2 ]; do
  MD_SO_OUT=$PWD/decompress_amd/libmdeflate_v$1.so MD_HIPCC_FLAGS="$2" python -m decompress_amd.build --force > /dev/null || exit 1
  echo built $1: $2
  shift 2
done
