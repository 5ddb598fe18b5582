#!/bin/bash
# round-3 decode-GEMM experiments (one GPU call): VALU issue rates incl. the fp8-pair converts, the dequant variants
# of the stripe kernel (variants/libdq1.so = v_cvt_pk_f32_fp8, bit-identical; libdq2.so = timing-only lower bound of a
# "scale after the MMA" conversion), and the K split / 2-workgroups-per-CU question on the gate_up shape.
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp1.txt
{
./scripts/ubench/valu_rate
echo "== dequant variants, M = 64 (scripts/bench_gemm.py, graph of 40 calls over 8 weight copies)"
bash scripts/ab_gemm.sh 64 default variants/libdq1.so variants/libdq2.so
echo "== dq1 exactness: w4a16 GPU tests on the variant"
MI355X_HOTPATH_LIB=$PWD/variants/libdq1.so timeout -k 10 600 python -m pytest tests/test_gpu_w4a16.py -x -q 2>&1 | tail -n 3
echo "== gate_up (plain, no SILU) with forced (nw, sk): 224 / 448 / 896 workgroups"
for f in 2,1 2,2 2,4 4,1 4,2 4,4; do echo "-- MI355X_STRIPE_FORCE=$f"; MI355X_STRIPE_FORCE=$f python scripts/bench_gemm.py 64 --only=gate_up; done
echo "== o_proj / down with forced sk"
for f in 2,2 2,4 2,8 2,16; do echo "-- MI355X_STRIPE_FORCE=$f"; MI355X_STRIPE_FORCE=$f python scripts/bench_gemm.py 64 --only=o --only=down --only=qkv; done
} > $O 2>&1
tail -n 60 $O
