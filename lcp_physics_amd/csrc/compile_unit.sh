#!/bin/bash
# compile_unit.sh SRC.hip OUT.o [compiler flags ...]  - one translation unit of liblcp_hip.so.
#
# hipcc -c in five explicit steps, with ONE extra step between the compiler and the assembler (round 6, profiles/r06_chain_rootcause.txt):
# ROCm 7.2's register allocator can place VGPR -> AGPR spill code at the top of the join block of a divergent `if`, IN FRONT of the
# `s_or_b64 exec, exec, sN` that re-enables the lanes - the spill then saves only the lanes that took the branch, the reload hands the
# others stale register contents (lcp_primal_kernel<56, ...>: wrong velocities; lcp_big_kernel<32, true, false> carried the same pattern
# unnoticed).  tools/isa_lint.py --fix moves such spill instructions behind the restore (a superset of the lanes: always safe) and refuses
# the build when it finds anything in such a shadow that it cannot move.  The device assembly stays in asm/ for tests/test_isa_lint.py.
#   1. device code to assembly      (hipcc --cuda-device-only -S: the same cc1 invocation as hipcc -c makes)
#   2. tools/isa_lint.py --fix
#   3. assemble, 4. link the code object, 5. bundle it      (the commands hipcc -### shows for the same steps)
#   6. host code with the bundle embedded                    (hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary)
set -e
SRC=$1; OUT=$2; shift 2
HERE=$(cd "$(dirname "$0")" && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
ARCH=${ARCH:-gfx950}
A=$HERE/asm
mkdir -p $A
B=$A/$(basename ${OUT%.o})
$HIPCC "$@" --offload-arch=$ARCH -x hip --cuda-device-only -S $SRC -o $B.s 2> $B.log || { cat $B.log >&2; exit 1; }
grep -v "argument unused during compilation: '--hip-link'" $B.log >&2 || true
python3 $HERE/../../tools/isa_lint.py --fix $B.s $B.fixed.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c $B.fixed.s -o $B.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B.hsaco $B.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH -input=/dev/null -input=$B.hsaco -output=$B.hipfb
$HIPCC "$@" --offload-arch=$ARCH -x hip --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B.hipfb -c $SRC -o $OUT
rm -f $B.dev.o $B.hsaco $B.hipfb $B.log
