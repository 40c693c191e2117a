#!/bin/bash
# tools/ab_off64.sh <outfile> -- (GPU box) what the 64-bit-gather-offset instantiations cost, and whether their 12-28 bytes of
# scratch at the 80-VGPR cap matter: (1) the SAME operand (products, X < 4 GiB) through the 32-bit and -- H2GCN_FORCE_OFF64 --
# the 64-bit kernels, 64- and 128-column slices, forward + adjoint; (2) the training step at the products shape (concat buffer
# > 4 GiB: the model dispatches the 64-bit kernels, incl. the accumulating general-store adjoint) with the default build
# (6 waves per SIMD, spills) against build/ab/lib_off64w5.so (-DH2GCN_OFF64_HEAVY_MIN_WAVES=5: 82-88 VGPRs, no scratch).
OUT=$1; : > $OUT
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), 'adjoint', round(d['adjoint']['kernel_ms'],3), round(d['adjoint']['frac'],4))"; }
for round in 1 2; do
 for lib in base off64w5; do
  for sl in 64 128; do
   for f in 0 1; do
    echo -n "round=$round lib=$lib slice=$sl force_off64=$f : " >> $OUT
    if [ $f = 1 ]; then export H2GCN_FORCE_OFF64=1; else unset H2GCN_FORCE_OFF64; fi
    H2GCN_HIP_LIBRARY=$ROOT/build/ab/lib_$lib.so timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --no-hbm-leg --steps 10 --warmup 3 --slice-cols $sl 2>/dev/null | tail -1 | line >> $OUT 2>&1
   done
  done
 done
done
unset H2GCN_FORCE_OFF64
for round in 1 2; do for lib in base off64w5; do for hidden in 64 100; do
  echo -n "round=$round lib=$lib train step hidden=$hidden : " >> $OUT
  H2GCN_HIP_LIBRARY=$ROOT/build/ab/lib_$lib.so timeout 900 python tools/epoch_products.py $hidden 2>/dev/null | tail -1 >> $OUT
done; done; done
cat $OUT
