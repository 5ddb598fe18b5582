// Micro-benchmark: do MFMAs and VALU work overlap on a gfx950 SIMD — inside one wave (interleaved
// instruction stream) and across the two waves of a SIMD (one wave MFMA-only, the other VALU-only)?
// Every wave runs `iters` rounds of 16 v_mfma_f32_16x16x32_bf16 (independent accumulators) and / or
// 64 v_pk_fma_f32 (8 independent chains); 256 workgroups x 512 threads = 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// the VALU instruction under test: -DVALU_OP=0 v_pk_fma_f32 (default), 1 v_fma_f32, 2 v_and_b32,
// 3 v_cvt_f32_ubyte0, 4 v_cvt_pk_bf16_f32
#ifndef VALU_OP
#define VALU_OP 0
#endif
#if VALU_OP == 0
#define VALU(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m), "v"(c))
#elif VALU_OP == 1
#define VALU(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r][0]) : "v"(m[0]), "v"(c[0]))
#elif VALU_OP == 2
#define VALU(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[r][0]) : "v"(m[0]))
#elif VALU_OP == 3
#define VALU(r) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(v[r][0]))
#else
#define VALU(r) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[r][0]) : "v"(m[0]))
#endif

// MODE 0: MFMA only; 1: VALU only; 2: both, interleaved 1 MFMA : 4 VALU in every wave;
// 3: waves 0-3 MFMA only (2x the rounds' MFMAs), waves 4-7 VALU only (2x the VALU): same total work
//    per SIMD as mode 2, split by wave;  4: both, all MFMAs then all VALU of a round (not interleaved)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)0.5f; }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  f32x2_t v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2_t{1.0f + i, 2.0f + threadIdx.x};
  const f32x2_t m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && wave >= 4);
  const int reps = MODE == 3 ? 2 : 1;
  for (int it = 0; it < iters * reps; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = (4 * i + j) & 7;
          VALU(r);
        }
      }
    } else {
      if (do_mfma) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int i = 0; i < 64; ++i)
          VALU(i & 7);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 4);
  const int iters = 40000;
  const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters),
              t3 = run<3>(d, iters), t4 = run<4>(d, iters);
  // cycles per round per SIMD at 2.4 GHz (2 waves per SIMD)
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / iters; };
  printf("MFMA only              : %7.2f ms  %6.0f cycles per round and SIMD (2 x 16 MFMAs)\n", t0, cyc(t0));
  printf("VALU only              : %7.2f ms  %6.0f cycles (2 x 64 VALU ops, -DVALU_OP)\n", t1, cyc(t1));
  printf("both, interleaved      : %7.2f ms  %6.0f cycles\n", t2, cyc(t2));
  printf("both, split by wave    : %7.2f ms  %6.0f cycles\n", t3, cyc(t3));
  printf("both, MFMAs then VALU  : %7.2f ms  %6.0f cycles\n", t4, cyc(t4));
  printf("sum %.0f, max %.0f\n", cyc(t0) + cyc(t1), cyc(t0) > cyc(t1) ? cyc(t0) : cyc(t1));
  return 0;
}
