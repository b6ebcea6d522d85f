#!/bin/bash
# Build A/B variants of liblcp_hip.so that differ in the -D flags of ONE translation unit (default lcp_quad.hip):
#   tools/build_variants.sh name1 "-DFLAG=1" name2 "-DA=0 -DB=1" ...      -> lcp_physics_amd/csrc/variants/<name>.so
# then on the GPU: bash tools/ab_bench.sh   (bench.py against every variant)
set -e
cd "$(dirname "$0")/../lcp_physics_amd/csrc"
UNIT=${UNIT:-lcp_quad}
mkdir -p variants
make -j8 > /dev/null
case $UNIT in lcp_quad*|lcp_solo) SLP=-fno-slp-vectorize;; *) SLP=;; esac      # (the Makefile's per-unit flag)
OTHERS=$(ls *.o | grep -v "^${UNIT}.o$" | grep -v "_prof.o$")
NAMES=""
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  NAMES="$NAMES $name"
  ( ./compile_unit.sh ${UNIT}.hip variants/${UNIT}_$name.o -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function $SLP $flags > variants/$name.log 2>&1 \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so $OTHERS variants/${UNIT}_$name.o && echo "built $name ($flags)" ) &
done
wait
rm -f variants/*.o
for name in $NAMES; do rm -f asm/${UNIT}_$name.s asm/${UNIT}_$name.fixed.s; done    # (only the variants' listings: asm/ also holds the product units')
