#!/bin/bash
# 8-bit decode GEMM weight stream: TIMING-ONLY variants with lane quads on contiguous 64-byte runs (results wrong)
cd "$(dirname "$0")/.."
O=gpurun_out/r03_exp18.txt
{
for lib in "" variants/libf8bc1.so variants/libf8bc2.so "" variants/libf8bc1.so variants/libf8bc2.so; do
  for which in 70b 8b; do
  echo "== lib=${lib:-default} $which"
  MI355X_HOTPATH_LIB=$lib timeout -k 10 200 python scripts/bench_scaled_mm_decode.py fp8 $which 2>&1 | grep "^fp8"
  done
done
} > $O 2>&1
grep -E "^==|total" $O
