#!/bin/bash
# usage: bash tools/kernel_resources.sh [object file] [name pattern]
# registers, LDS and scratch of the gfx950 kernels in an object file of the library (default: build/tscm_solver.o), from the
# code object's metadata notes
o=${1:-$(dirname $0)/../tscm_calib_amd/csrc/build/tscm_solver.o}; pat=${2:-.}
t=$(mktemp -d); trap 'rm -rf $t' EXIT
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin $o
$B/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/k.co
$B/llvm-readelf --notes $t/k.co | python3 -c "
import re, sys, subprocess
txt = sys.stdin.read()
for blk in re.split(r'\n  - \.agpr_count', txt)[1:]:
    blk = '.agpr_count' + blk
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(.*', '', name).replace('void ', '').replace('tscm::', '')
    if re.search(r'''$pat''', name):
        print(f\"{name:48s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s}  lds {g('group_segment_fixed_size'):>6s}  scratch {g('private_segment_fixed_size'):>5s}  spills v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}\")
"
