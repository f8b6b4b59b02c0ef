#!/bin/bash
# Run on the GPU box: the long parity soaks of the final build (bit-exact families, tolerance families, sequences through the fused frame path).
set -u
OUT=gpurun_out/soak_final
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
cat libcml_amd/BUILD_COMMIT > $OUT/BUILD_COMMIT 2>/dev/null
N=${1:-120}
NS=${2:-120}
( time timeout 1500 python tests/soak_parity.py $N ) > $OUT/soak.txt 2>&1
( time timeout 900 python tests/soak_parity.py $N --tolerance ) > $OUT/soak_tolerance.txt 2>&1
( time timeout 1800 python tests/soak_parity.py $NS --sequence ) > $OUT/soak_sequence.txt 2>&1
tail -15 $OUT/soak.txt; tail -25 $OUT/soak_tolerance.txt; tail -12 $OUT/soak_sequence.txt
