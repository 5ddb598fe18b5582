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
