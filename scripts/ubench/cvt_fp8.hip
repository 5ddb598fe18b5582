// cvt_fp8.hip — what do gfx950's fp8 -> 16-bit conversions compute?  (semantics probe for the fp8 KV path)
//   v_cvt_scalef32_pk_bf16_fp8 / _pk_f16_fp8 with scale 1.0, 0.5 (power of two) and 0.3 (not one),
//   against v_cvt_pk_f32_fp8 (exact) * scale rounded to the 16-bit type on the host.
// build: hipcc --offload-arch=gfx950 -O2 cvt_fp8.hip -o cvt_fp8 ; prints one line per (type, scale).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__global__ void probe(bf16x2_t* ob, f16x2_t* oh, f32x2_t* of, float s) {
  const unsigned v = threadIdx.x | ((255u - threadIdx.x) << 8);   // byte 0 = code, byte 1 = 255 - code
  ob[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, s, false);
  oh[threadIdx.x] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v, s, false);
  of[threadIdx.x] = __builtin_amdgcn_cvt_pk_f32_fp8(v, false);
}

static float bf16_rne(float x) {
  unsigned u; memcpy(&u, &x, 4);
  if (std::isnan(x)) return x;
  u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;
  float r; memcpy(&r, &u, 4); return r;
}

int main() {
  bf16x2_t* ob; f16x2_t* oh; f32x2_t* of;
  hipMalloc(&ob, 256 * sizeof(bf16x2_t)); hipMalloc(&oh, 256 * sizeof(f16x2_t)); hipMalloc(&of, 256 * sizeof(f32x2_t));
  for (float s : {1.0f, 0.5f, 0.3f, 3.0f}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, ob, oh, of, s);
    std::vector<bf16x2_t> hb(256); std::vector<f16x2_t> hh(256); std::vector<f32x2_t> hf(256);
    hipMemcpy(hb.data(), ob, 256 * sizeof(bf16x2_t), hipMemcpyDeviceToHost);
    hipMemcpy(hh.data(), oh, 256 * sizeof(f16x2_t), hipMemcpyDeviceToHost);
    hipMemcpy(hf.data(), of, 256 * sizeof(f32x2_t), hipMemcpyDeviceToHost);
    int bad_mul_b = 0, bad_exp_b = 0, bad_mul_h = 0, nan = 0;
    int e; std::frexp(s, &e); const float s_pow2 = std::ldexp(1.0f, e - 1);   // exponent of s only
    for (int c = 0; c < 256; ++c) {
      const float f = hf[c][0];
      if (std::isnan(f)) { ++nan; continue; }
      const float got_b = (float)hb[c][0], got_h = (float)hh[c][0];
      if (got_b != bf16_rne(f * s)) ++bad_mul_b;
      if (got_b != bf16_rne(f * s_pow2)) ++bad_exp_b;
      if (got_h != (float)(_Float16)(f * s)) ++bad_mul_h;
    }
    printf("scale %.2f: bf16 != rne(f*scale): %d, bf16 != rne(f*2^floor(log2 scale)): %d, f16 != rne(f*scale): %d (NaN codes %d); "
           "code 0x38 -> f32 %.4f bf16 %.4f\n", s, bad_mul_b, bad_exp_b, bad_mul_h, nan, hf[0x38][0], (float)hb[0x38][0]);
  }
  return 0;
}
