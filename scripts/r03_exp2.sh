#!/bin/bash
# round-3 decode-GEMM experiments, call 2: in-kernel stamps of the default / halved-VALU (dq2) / deeper-weight-ring
# builds, then the ring-depth variants on the bench shapes.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp2.txt
{
for v in stamp0 stampdq2 stampwd5 stampdq2wd5; do
  echo "== stamps: $v (gate_up 28672 x 4096, then o_proj 4096 x 4096)"
  MI355X_HOTPATH_LIB=$PWD/variants/lib$v.so python scripts/stamp_stripe.py 2>&1 | grep -v amdgpu | head -n 12
done
echo "== ring depth variants, M = 64"
bash scripts/ab_gemm.sh 64 default variants/libwd4.so variants/libwd5.so variants/libwd5ad4.so variants/libdq2wd5.so 2>&1 | grep -v amdgpu
} > $O 2>&1
tail -n 80 $O
