#!/bin/bash
# VGPRs / SGPRs / LDS / scratch of every kernel in the built library (from the code object's metadata)
cd "$(dirname "$0")/../.."
SO=${1:-decompress_amd/libmdeflate.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$SO >/dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section=.hip_fatbin=$T/fat.bin $SO 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co 2>/dev/null || \
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)', txt, re.S):
    print('%-90s lds %6s scratch %4s sgpr %3s vgpr %3s' % (m.group(2)[:90], m.group(1), m.group(3), m.group(4), m.group(5)))
"
rm -rf $T
