#!/bin/bash
# Re-assemble a (hand-edited) device .s of ONE translation unit into a liblcp_hip variant, without letting the compiler near it again:
#   tools/asm_rebuild.sh <tempdir of `hipcc -save-temps` with cmds.txt> <edited device .s> <unit object name, e.g. lcp_primal_chain> <out.so>
# cmds.txt = the `hipcc -### -save-temps` command list of that unit (lines 4.. = device assembler, lld, bundler, host compile with the new fat binary).
# Used by the round-6 root-cause work (profiles/r06_chain_rootcause.txt): register dumps inserted at the assembly level leave register
# allocation and scheduling of the failing build untouched.
set -e
D=$1; S=$2; UNIT=$3; OUT=$4
CS=$(cd "$(dirname "$0")/../lcp_physics_amd/csrc" && pwd)
cd $D
cp $S lcp_primal_chain-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null || true
for n in 4 5 6 8 9 10; do
  sed -n "${n}p" cmds.txt > /tmp/_cmd.sh
  bash /tmp/_cmd.sh
done
OTHERS=$(ls $CS/*.o | grep -v "/${UNIT}.o$" | grep -v "_prof.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $OTHERS $D/chain.o
echo "built $OUT"
