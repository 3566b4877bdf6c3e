#!/bin/bash
# timing lab: which block type carries the duration of the two small kernels (CA_LAB_SKIP bits; results are wrong on purpose)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for args in "--cells 10000 --genes 2000 --clones 4" "--cells 12500" ""; do
  echo "== $args"
  bash $ROOT/tools/lab_ab.sh "$args" mon="-DCA_LAB_SKIP=1" psi="-DCA_LAB_SKIP=2" gene="-DCA_LAB_SKIP=4" fg_all="-DCA_LAB_SKIP=7" small="-DCA_LAB_SKIP=8" pre="-DCA_LAB_SKIP=16" ysq="-DCA_LAB_SKIP=32" cell="-DCA_LAB_SKIP=64" ac_all="-DCA_LAB_SKIP=120"
done
