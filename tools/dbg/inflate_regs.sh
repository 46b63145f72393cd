#!/bin/bash
# compile inflate_wave.hip alone (device code only) and print its kernels' resources; extra args go to hipcc
cd "$(dirname "$0")/../.."
SRC=${SRC:-decompress_amd/csrc/inflate_wave.hip}
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -falign-loops=64 -Wno-unused-value -I include -I decompress_amd/csrc --cuda-device-only --no-gpu-bundle-output -c "$@" -o $T/k.co $SRC || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)', txt, re.S):
    print('%-60s lds %6s scratch %4s sgpr %3s vgpr %3s' % (m.group(2)[:60], m.group(1), m.group(3), m.group(4), m.group(5)))
"
[ -n "$KEEP" ] && cp $T/k.co $KEEP
rm -rf $T
