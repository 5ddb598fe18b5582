// Micro-benchmark: sustained v_mfma_f32_16x16x32_bf16 rate with operands in registers (no memory
// traffic), all CUs busy: the practical MFMA ceiling under the chip's power / clock management.
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  // RANDOM: 8 different pseudo-random operand pairs cycled through (realistic bit toggling in the
  // multiplier arrays); otherwise one constant pair
  bf16x8_t av[8], bv[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int r = 0; r < 8; ++r)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      const float x = RANDOM ? ((int)(h >> 9) - (1 << 22)) * (1.0f / (1 << 22)) : 1.0f;
      h = h * 1664525u + 1013904223u;
      const float y = RANDOM ? ((int)(h >> 9) - (1 << 22)) * (1.0f / (1 << 22)) : 0.5f;
      av[r][i] = (__bf16)x; bv[r][i] = (__bf16)y;
    }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 7], bv[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rnd = 0; rnd <= 1; ++rnd)
  for (int waves = 2; waves <= 2; ++waves) {
    for (int ms_target : {20, 200}) {
      const int iters = ms_target * 9000;   // ~ms_target ms of MFMAs
      float ms = 0;
      for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        if (rnd) hipLaunchKernelGGL(k<true>, dim3(256), dim3(256 * waves), 0, 0, d, iters);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(256 * waves), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = 256.0 * (4 * waves) * iters * 16.0 * 16384.0;
      printf("%s operands, %d wave(s)/SIMD, %7.1f ms: %.0f TFLOP/s\n", rnd ? "random" : "constant", waves, ms, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
