// activation.hip — SwiGLU gate: out[t, i] = T(silu(x)) * y, x = in[t, i], y = in[t, d+i].
//
// Reference restated: csrc/activation_kernels.cu:14-36 (compute / kernel),
// :142-147 (silu_kernel: T( x / (1 + expf(-x)) ) in fp32, rounded to T, then the
// T*T product is rounded again), :177-247 (launch).  2-D grid over (column tile,
// token) so that small decode batches still fill the chip; 16 B per lane.
#include "common.cuh"

namespace mi355x {

// QUANT: the product is additionally quantised to fp8 e4m3 with a static per-tensor scale,
// fp8(T(silu(x)) * y * (1 / scale)) saturated to +-448 (csrc/quantization/activation_kernels.cu:21-90:
// the T*T product is rounded to T first, then scaled_fp8_conversion<true>).
template <typename T, bool VEC, bool QUANT>
__global__ void silu_and_mul_kernel(void* __restrict__ out_, const T* __restrict__ in, int d,
                                    const float* __restrict__ scale) {
  const int64_t token = blockIdx.x;
  const int bx = blockIdx.y;
  float inv = 1.f;
  if constexpr (QUANT) inv = 1.f / *scale;
  if constexpr (QUANT) {
    const T* x = in + token * 2 * d;
    const T* y = x + d;
    uint8_t* o = static_cast<uint8_t*>(out_) + token * d;
    if constexpr (VEC) {
      constexpr int V = 16 / sizeof(T);
      const int i = (bx * blockDim.x + threadIdx.x) * V;
      if (i >= d) return;
      const Vec16<T> xv = load16(x + i);
      const Vec16<T> yv = load16(y + i);
      uint8_t q[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        q[j] = f32_to_fp8_sat(to_f32(mul_t<T>(silu_t<T>(xv.e[j]), yv.e[j])) * inv);
      }
      if constexpr (V == 8) *reinterpret_cast<uint2*>(o + i) = *reinterpret_cast<const uint2*>(q);
      else *reinterpret_cast<uint32_t*>(o + i) = *reinterpret_cast<const uint32_t*>(q);
    } else {
      const int i = bx * blockDim.x + threadIdx.x;
      if (i >= d) return;
      o[i] = f32_to_fp8_sat(to_f32(mul_t<T>(silu_t<T>(x[i]), y[i])) * inv);
    }
    return;
  }
  T* out = static_cast<T*>(out_);
  const T* x = in + token * 2 * d;
  const T* y = x + d;
  T* o = out + token * d;
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    const int i = (bx * blockDim.x + threadIdx.x) * V;
    if (i >= d) return;
    const Vec16<T> xv = load16(x + i);
    const Vec16<T> yv = load16(y + i);
    Vec16<T> r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.e[j] = mul_t<T>(silu_t<T>(xv.e[j]), yv.e[j]);
    store16(o + i, r);
  } else {
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= d) return;
    o[i] = mul_t<T>(silu_t<T>(x[i]), y[i]);
  }
}

// silu_and_mul + dynamic per-token fp8 quantisation in one launch (the input of an fp8 down_proj): one workgroup
// per token, the T-rounded products stay in registers between the row maximum and the quantisation.  The bits of
// silu_and_mul followed by dynamic_per_token_scaled_fp8_quant (csrc/activation_kernels.cu:14-36, csrc/quantization/
// fp8/common.cu:91-133: s = max(absmax / 448, 1 / (448 * 512)), q = sat(float(x) / s)); no reference op of its own.
// SLABS: the gate_up row is still the split-K partial slabs [sk][tokens][2 d] of an fp8 GEMM
// (mi355x_scaled_mm_fp8_deferred): x = T(sum * a_scale[token] * b_scale[column]) = what its finish kernel stores.
// NT threads per workgroup: 256, or 1024 for decode-sized batches (one workgroup per token: with 64 tokens only 64
// CUs work, four times the loads in flight per CU then: 8B gate_up slabs 26.5 -> see profiles/r03 notes)
template <typename T, int MAXC, bool SLABS = false, int NT = 256>
__global__ __launch_bounds__(NT) void silu_mul_per_token_quant_kernel(
    uint8_t* __restrict__ out, float* __restrict__ scales, const T* __restrict__ in, int d,
    const float* __restrict__ slabs = nullptr, int sk = 0, int64_t slab_stride = 0,
    SlabScales slab_scales = SlabScales{nullptr, nullptr, 0, 0}) {
  __shared__ float red[16];
  constexpr int V = 16 / sizeof(T);
  const int64_t token = blockIdx.x;
  const T* x = in + token * 2 * d;
  const T* y = x + d;
  Vec16<T> act[MAXC];
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * NT + threadIdx.x) * V;
    if (i < d) {
      Vec16<T> xv, yv;
      if constexpr (SLABS) {
        const float* sp = slabs + token * 2 * d;
        slab_values<T, V>(sp + i, sk, slab_stride, slab_scales, token, i, xv.e);
        slab_values<T, V>(sp + d + i, sk, slab_stride, slab_scales, token, d + i, yv.e);
      } else {
        xv = load16(x + i);
        yv = load16(y + i);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        act[c].e[j] = mul_t<T>(silu_t<T>(xv.e[j]), yv.e[j]);
        m = fmaxf(m, fabsf(to_f32(act[c].e[j])));
      }
    }
  }
  m = block_reduce<true>(m, red);
  const float s = fmaxf(m / kFp8Max, kFp8MinScale);
  if (threadIdx.x == 0) scales[token] = s;
  const RowDiv rdiv = make_row_div(s);
  uint8_t* o = out + token * d;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int i = (c * NT + threadIdx.x) * V;
    if (i < d) {
      float y[V];
#pragma unroll
      for (int j = 0; j < V; ++j) y[j] = row_div(to_f32(act[c].e[j]), rdiv);
      *reinterpret_cast<uint2*>(o + i) = f32x8_to_fp8x8_sat(y);
    }
  }
}

}  // namespace mi355x

using namespace mi355x;

template <bool QUANT>
static int launch_silu(void* out, const void* input, const float* scale, int num_tokens, int d,
                       int dtype, hipStream_t s, const char* what) {
  return MI355X_DISPATCH_FLOAT(dtype, [&]() -> int {
    constexpr int V = 16 / sizeof(scalar_t);
    const scalar_t* in = static_cast<const scalar_t*>(input);
    // 16-B loads; the stores are 16 B (T) or V bytes (fp8)
    const uintptr_t out_mask = QUANT ? V - 1 : 15;
    const bool vec = d % V == 0 && (reinterpret_cast<uintptr_t>(out) & out_mask) == 0 &&
                     (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const int work = vec ? d / V : d;
    const int threads = work >= 256 ? 256 : ((work + 63) / 64) * 64;
    const int col_blocks = (work + threads - 1) / threads;
    if (col_blocks > 65535) {
      set_error("%s: d = %d too large", what, d);
      return MI355X_EUNSUPPORTED;
    }
    dim3 grid(num_tokens, col_blocks), block(threads);
    if (vec)
      hipLaunchKernelGGL((silu_and_mul_kernel<scalar_t, true, QUANT>), grid, block, 0, s, out, in, d,
                         scale);
    else
      hipLaunchKernelGGL((silu_and_mul_kernel<scalar_t, false, QUANT>), grid, block, 0, s, out, in, d,
                         scale);
    return check_launch(what);
  });
}

extern "C" int mi355x_silu_and_mul(void* out, const void* input, int num_tokens, int d,
                                   int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && d > 0, MI355X_EINVAL, "silu_and_mul: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input, MI355X_EINVAL, "silu_and_mul: null pointer");
  return launch_silu<false>(out, input, nullptr, num_tokens, d, dtype,
                            static_cast<hipStream_t>(stream), "silu_and_mul");
}

extern "C" int mi355x_silu_and_mul_quant(void* out, const void* input, const float* scale,
                                         int num_tokens, int d, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && d > 0, MI355X_EINVAL, "silu_and_mul_quant: bad sizes");
  MI355X_REQUIRE(dtype == MI355X_F16 || dtype == MI355X_BF16, MI355X_EUNSUPPORTED,
                 "silu_and_mul_quant: input must be fp16 or bf16");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scale, MI355X_EINVAL, "silu_and_mul_quant: null pointer");
  return launch_silu<true>(out, input, scale, num_tokens, d, dtype,
                           static_cast<hipStream_t>(stream), "silu_and_mul_quant");
}

// returns 1 (no error) when the fused form does not apply (d % 8 != 0, d > 16384, unaligned): run the two ops
extern "C" int mi355x_silu_and_mul_per_token_quant(void* out, float* scales, const void* input, int num_tokens,
                                                   int d, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && d > 0, MI355X_EINVAL, "silu_and_mul_per_token_quant: bad sizes");
  MI355X_REQUIRE(dtype == MI355X_F16 || dtype == MI355X_BF16, MI355X_EUNSUPPORTED,
                 "silu_and_mul_per_token_quant: input must be fp16 or bf16");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && scales && input, MI355X_EINVAL, "silu_and_mul_per_token_quant: null pointer");
  if (d % 8 != 0 || d > 16384 || (reinterpret_cast<uintptr_t>(input) & 15) || (reinterpret_cast<uintptr_t>(out) & 7))
    return 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_HALF(dtype, [&]() -> int {
    const scalar_t* in = static_cast<const scalar_t*>(input);
    uint8_t* o = static_cast<uint8_t*>(out);
    if (num_tokens <= 128 && d > 2048) {   // decode: 1024 threads per token
      const int chunks = (d / 8 + 1023) / 1024;
      if (chunks <= 1)
        hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 1, false, 1024>), dim3(num_tokens), dim3(1024), 0, s, o, scales, in, d);
      else
        hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 2, false, 1024>), dim3(num_tokens), dim3(1024), 0, s, o, scales, in, d);
      return check_launch("silu_and_mul_per_token_quant");
    }
    const int chunks = (d / 8 + 255) / 256;
    if (chunks <= 2)
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 2>), dim3(num_tokens), dim3(256), 0, s, o, scales, in, d);
    else if (chunks <= 4)
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 4>), dim3(num_tokens), dim3(256), 0, s, o, scales, in, d);
    else
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 8>), dim3(num_tokens), dim3(256), 0, s, o, scales, in, d);
    return check_launch("silu_and_mul_per_token_quant");
  });
}

// The same with the gate_up row still in the split-K slabs of an fp8 GEMM (mi355x_scaled_mm_fp8_deferred);
// bit-identical to the GEMM's finish launch followed by the op above.  Returns 1 when not applicable.
extern "C" int mi355x_silu_and_mul_per_token_quant_slabs(void* out, float* scales, const float* slabs, int sk,
                                                         const float* a_scales, int a_scales_numel,
                                                         const float* b_scales, int b_scales_numel,
                                                         int num_tokens, int d, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && d > 0 && sk > 0, MI355X_EINVAL, "silu_and_mul_per_token_quant_slabs: bad sizes");
  MI355X_REQUIRE(dtype == MI355X_F16 || dtype == MI355X_BF16, MI355X_EUNSUPPORTED,
                 "silu_and_mul_per_token_quant_slabs: the GEMM's output type must be fp16 or bf16");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && scales && slabs && a_scales && b_scales, MI355X_EINVAL,
                 "silu_and_mul_per_token_quant_slabs: null pointer");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == num_tokens) &&
                     (b_scales_numel == 1 || b_scales_numel == 2 * d),
                 MI355X_EINVAL, "silu_and_mul_per_token_quant_slabs: scales per-tensor or per-token / per-column");
  if (d % 8 != 0 || d > 16384 || (reinterpret_cast<uintptr_t>(slabs) & 15) || (reinterpret_cast<uintptr_t>(out) & 7))
    return 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const SlabScales sc{a_scales, b_scales, a_scales_numel > 1, b_scales_numel > 1};
  const int64_t stride = (int64_t)num_tokens * 2 * d;
  return MI355X_DISPATCH_HALF(dtype, [&]() -> int {
    uint8_t* o = static_cast<uint8_t*>(out);
    const scalar_t* none = nullptr;
    if (num_tokens <= 128 && d > 2048) {   // decode: 1024 threads per token
      const int chunks = (d / 8 + 1023) / 1024;
      if (chunks <= 1)
        hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 1, true, 1024>), dim3(num_tokens), dim3(1024), 0,
                           s, o, scales, none, d, slabs, sk, stride, sc);
      else
        hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 2, true, 1024>), dim3(num_tokens), dim3(1024), 0,
                           s, o, scales, none, d, slabs, sk, stride, sc);
      return check_launch("silu_and_mul_per_token_quant_slabs");
    }
    const int chunks = (d / 8 + 255) / 256;
    if (chunks <= 2)
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 2, true>), dim3(num_tokens), dim3(256), 0, s, o,
                         scales, none, d, slabs, sk, stride, sc);
    else if (chunks <= 4)
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 4, true>), dim3(num_tokens), dim3(256), 0, s, o,
                         scales, none, d, slabs, sk, stride, sc);
    else
      hipLaunchKernelGGL((silu_mul_per_token_quant_kernel<scalar_t, 8, true>), dim3(num_tokens), dim3(256), 0, s, o,
                         scales, none, d, slabs, sk, stride, sc);
    return check_launch("silu_and_mul_per_token_quant_slabs");
  });
}
