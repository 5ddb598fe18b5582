for r in 1 2; do
for v in 0 1; do echo "== MI355X_F8_NT=$v"; MI355X_F8_NT=$v python scripts/bench_scaled_mm_decode.py fp8 70b 2>&1 | grep -v amdgpu; done
done
for r in 1 2; do
for v in 0 1; do echo "== MI355X_GEMM_4W=$v"; MI355X_GEMM_4W=$v python scripts/bench_gemm_pp.py 2>&1 | grep -v amdgpu; done
done
MI355X_GEMM_4W=1 python -m pytest tests/test_gpu_w4a16.py -x -q 2>&1 | tail -3
