#!/bin/bash
# VGPR / spill counts of the device kernels of one translation unit: scripts/kernel_regs.sh engine.hip [name filter]
# (device-only compile with the Makefile's flags, unbundle the gfx950 code object, read the kernel descriptors' metadata)
set -e
src=${1:-engine.hip}; filt=${2:-}
cd "$(dirname "$0")/../ecad_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -ffast-math -fno-finite-math-only --cuda-device-only -c "$src" -o "$tmp/dev.bundle"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$tmp/dev.bundle" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$tmp/dev.elf"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.elf" | python3 -c "
import re, sys
for blk in sys.stdin.read().split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if '$filt' in name:
        f = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)
        print(f'{name[:120]:120s} vgpr {f(\"vgpr_count\"):>3s} spill {f(\"vgpr_spill_count\"):>3s} lds {f(\"group_segment_fixed_size\")}')
" | sort
rm -rf "$tmp"
