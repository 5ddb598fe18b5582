// Micro-benchmark: issue rate of the VALU instructions the w4a16 dequant uses (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int WHICH>
__global__ void k(uint64_t* out, uint32_t seed, int iters) {
  uint32_t a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
  float f0 = a, f1 = b, f2 = c, f3 = d, g0 = 1.5f, g1 = 0.25f;
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t p0 = {f0, f1}, p1 = {f2, f3}, p2 = {g0, g1}, p3 = {g1, g0};
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (WHICH == 0) { REP16(asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f0) : "v"(a));
                       asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f1) : "v"(b));
                       asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(f2) : "v"(c));
                       asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f3) : "v"(d));) }
    if (WHICH == 1) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(f0), "v"(f1));
                       asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b) : "v"(f2), "v"(f3));
                       asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c) : "v"(f1), "v"(f2));
                       asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(f3), "v"(f0));) }
    if (WHICH == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p0) : "v"(p1), "v"(p2), "v"(p3));
                       asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p1) : "v"(p2), "v"(p3), "v"(p2));
                       asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p0) : "v"(p1), "v"(p2), "v"(p3));
                       asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p1) : "v"(p2), "v"(p3), "v"(p2));) }
    if (WHICH == 3) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(f1), "v"(g0), "v"(g1));
                       asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(f2), "v"(g0), "v"(g1));
                       asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f2) : "v"(f3), "v"(g0), "v"(g1));
                       asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(f0), "v"(g0), "v"(g1));) }
    if (WHICH == 4) { REP16(asm volatile("v_and_b32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
                       asm volatile("v_lshrrev_b32 %0, 4, %1" : "=v"(b) : "v"(c));
                       asm volatile("v_and_b32 %0, %1, %2" : "=v"(c) : "v"(d), "v"(a));
                       asm volatile("v_bfe_u32 %0, %1, 4, 4" : "=v"(d) : "v"(a));) }
    if (WHICH == 5) { REP16(asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "s"(seed));
                       asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b) : "v"(c), "v"(d), "s"(seed));
                       asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "v"(a), "s"(seed));
                       asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(seed));) }
    if (WHICH == 6) { REP16(asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "s"(seed), "v"(c));
                       asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(b) : "v"(c), "v"(d));
                       asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(c) : "v"(d), "s"(seed), "v"(a));
                       asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(a), "v"(b));) }
    if (WHICH == 7) { REP16(asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f0) : "v"(a));
                       asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f1) : "v"(b));
                       asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(f2) : "v"(c));
                       asm volatile("v_cvt_f32_i32_e32 %0, %1" : "=v"(f3) : "v"(d));) }
    // round 3: fp8 -> f32 pair converts (a byte 0x0q read as e4m3 is q * 2^-9: the int4 nibble as a denormal-continuous fp8)
    if (WHICH == 8) { REP16(asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_0" : "=v"(p0) : "v"(a));
                       asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(p1) : "v"(b));
                       asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_0" : "=v"(p2) : "v"(c));
                       asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(p3) : "v"(d));) }
    if (WHICH == 9) { REP16(asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(p0) : "v"(a), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(p1) : "v"(b), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(p2) : "v"(c), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(p3) : "v"(d), "v"(g0));) }
    if (WHICH == 10) { REP16(asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(a) : "v"(b), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2 op_sel:[1,0,0,0]" : "=v"(b) : "v"(c), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(c) : "v"(d), "v"(g0));
                       asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(g0));) }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (f0 + f1 + f2 + f3 + p0.x + p1.y + p2.x + p3.y == 12345.f && a + b + c + d == 77) out[1] = 1;
}
__global__ void nib(float* out) {
  // out[q] = v_cvt_pk_f32_fp8(0x0q) for q = 0..15 (expected q * 2^-9 with OCP e4m3)
  typedef float f2_t __attribute__((ext_vector_type(2)));
  uint32_t w = threadIdx.x & 15;
  f2_t r;
  asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_0" : "=v"(r) : "v"(w));
  if (threadIdx.x < 16) out[threadIdx.x] = r.x;
}
int main() {
  { float* o; hipMalloc(&o, 64); hipLaunchKernelGGL(nib, dim3(1), dim3(64), 0, 0, o); float h[16];
    hipMemcpy(h, o, 64, hipMemcpyDeviceToHost); printf("nibble as e4m3 * 512:");
    for (int i = 0; i < 16; ++i) printf(" %g", h[i] * 512.f); printf("\n"); }
  uint64_t* d; hipMalloc(&d, 16);
  const char* names[] = {"v_cvt_f32_ubyteN", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "v_fma_f32", "and/shift/bfe", "v_perm_b32", "v_and_or / v_lshl_or", "v_cvt_f32_u32 (sdwa/plain)", "v_cvt_pk_f32_fp8 (sdwa)",
                         "v_cvt_scalef32_pk_f32_fp8", "v_cvt_scalef32_pk_bf16_fp8"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 1; waves <= 4; waves *= 2) {
    for (int w = 0; w < 11; ++w) {
      const int iters = 20000;
      dim3 block(256 * waves), grid(256);  // one block per CU, `waves` waves per SIMD
      #define L(W) hipLaunchKernelGGL(k<W>, grid, block, 0, 0, d, 1u, iters)
      float ms = 0;
      for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        if (w == 0) L(0); if (w == 1) L(1); if (w == 2) L(2); if (w == 3) L(3); if (w == 4) L(4); if (w == 5) L(5); if (w == 6) L(6); if (w == 7) L(7); if (w == 8) L(8); if (w == 9) L(9); if (w == 10) L(10);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      // SIMD-cycles per wave-instruction at 2.4 GHz: time * 2.4e9 / (instructions per SIMD)
      const double per_simd = (double)iters * 64.0 * waves;
      printf("%d wave(s)/SIMD  %-20s %.2f cycles (2.4 GHz) per wave-instruction per SIMD\n", waves, names[w],
             ms * 1e-3 * 2.4e9 / per_simd);
    }
  }
  return 0;
}
