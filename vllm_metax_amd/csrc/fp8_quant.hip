// fp8_quant.hip — activation -> float8_e4m3fn quantisation for gfx950.
//
// Reference semantics restated (csrc/quantization/fp8/common.cu:11-133,
// common.cuh:25-38, quantization/utils.cuh:15-44):
//   static :  q = sat( float(x) * (1/scale) )
//   dynamic:  scale = max over tensor of absmax(row)/448, folded with an atomic max
//             into a caller-zeroed scalar; then q = sat( float(x) * (1/scale) )
//   per-tok:  s = max( min(absmax, ub)/448, 1/(448*512) ); q = sat( float(x) / s )
// sat() clamps to +-448 and converts RNE (v_cvt_pk_fp8_f32, OCP e4m3fn on gfx950).
// One workgroup per token row, 16-B loads / 8-B stores when the row is aligned.
#include "common.cuh"

namespace mi355x {

template <typename T, bool VEC, typename F>
__device__ __forceinline__ void for_each_row_elem(const T* row, int n, F&& f) {
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    for (int i = threadIdx.x * V; i < n; i += blockDim.x * V) {
      T v[V];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(row + i);
#pragma unroll
      for (int j = 0; j < V; ++j) f(i + j, to_f32(v[j]));
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) f(i, to_f32(row[i]));
  }
}

template <typename T, bool VEC, bool DIVIDE>
__device__ __forceinline__ void quant_row(uint8_t* out_row, const T* row, int n, float s) {
  RowDiv rdiv{};
  if constexpr (DIVIDE) rdiv = make_row_div(s);   // x / s, the bits of the division (common.cuh)
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    for (int i = threadIdx.x * V; i < n; i += blockDim.x * V) {
      T v[V];
      float y[V];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(row + i);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float x = to_f32(v[j]);
        if constexpr (DIVIDE) y[j] = row_div(x, rdiv);
        else y[j] = x * s;
      }
      if constexpr (V == 8) {
        *reinterpret_cast<uint2*>(out_row + i) = f32x8_to_fp8x8_sat(y);
      } else {
        *reinterpret_cast<uint32_t*>(out_row + i) = f32x4_to_fp8x4_sat(y);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float x = to_f32(row[i]);
      float y;
      if constexpr (DIVIDE) y = row_div(x, rdiv);
      else y = x * s;
      out_row[i] = f32_to_fp8_sat(y);
    }
  }
}

// static and dynamic-per-tensor second pass: multiply by 1/scale
template <typename T, bool VEC>
__global__ void scaled_fp8_quant_kernel(uint8_t* __restrict__ out, const T* __restrict__ in,
                                        const float* __restrict__ scale, int hidden,
                                        int64_t in_stride, int64_t out_stride) {
  const int64_t row = blockIdx.x;
  const float inv = 1.0f / *scale;
  quant_row<T, VEC, false>(out + row * out_stride, in + row * in_stride, hidden, inv);
}

template <typename T, bool VEC>
__global__ void absmax_to_scale_kernel(float* __restrict__ scale, const T* __restrict__ in,
                                       int hidden, int64_t in_stride) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  float m = 0.f;
  for_each_row_elem<T, VEC>(in + row * in_stride, hidden,
                            [&](int, float v) { m = fmaxf(m, fabsf(v)); });
  m = block_reduce<true>(m, red);
  if (threadIdx.x == 0) {
    // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<int*>(scale), __float_as_int(m / kFp8Max));
  }
}

template <typename T, bool VEC>
__global__ void per_token_fp8_quant_kernel(uint8_t* __restrict__ out,
                                           float* __restrict__ scales,
                                           const T* __restrict__ in,
                                           const float* __restrict__ scale_ub, int hidden,
                                           int64_t in_stride, int64_t out_stride) {
  __shared__ float red[16];
  __shared__ float s_scale;
  const int64_t row = blockIdx.x;
  const T* rin = in + row * in_stride;
  float m = 0.f;
  for_each_row_elem<T, VEC>(rin, hidden, [&](int, float v) { m = fmaxf(m, fabsf(v)); });
  m = block_reduce<true>(m, red);
  if (threadIdx.x == 0) {
    float s = scale_ub ? fminf(m, *scale_ub) : m;
    s = fmaxf(s / kFp8Max, kFp8MinScale);
    scales[row] = s;
    s_scale = s;
  }
  __syncthreads();
  quant_row<T, VEC, true>(out + row * out_stride, rin, hidden, s_scale);
}

template <typename T>
static bool row_vec_ok(const void* in, const void* out, int hidden, int64_t in_stride,
                       int64_t out_stride) {
  constexpr int V = 16 / sizeof(T);
  return hidden % V == 0 && in_stride % V == 0 && out_stride % V == 0 &&
         (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(out) % V) == 0;
}

static int row_threads(int hidden, int per_thread) {
  int t = (hidden + per_thread - 1) / per_thread;
  t = ((t + 63) / 64) * 64;
  return t > 256 ? 256 : t;
}

}  // namespace mi355x

using namespace mi355x;

extern "C" {

int mi355x_static_scaled_fp8_quant(void* out, const void* input, const float* scale,
                                   int num_tokens, int hidden_size,
                                   int64_t in_row_stride, int64_t out_row_stride,
                                   int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "static_scaled_fp8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scale, MI355X_EINVAL, "static_scaled_fp8_quant: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    const scalar_t* in = static_cast<const scalar_t*>(input);
    uint8_t* o = static_cast<uint8_t*>(out);
    const bool vec = row_vec_ok<scalar_t>(in, o, hidden_size, in_row_stride, out_row_stride);
    dim3 grid(num_tokens), block(row_threads(hidden_size, vec ? 16 / sizeof(scalar_t) : 1));
    if (vec)
      hipLaunchKernelGGL((scaled_fp8_quant_kernel<scalar_t, true>), grid, block, 0, s, o, in,
                         scale, hidden_size, in_row_stride, out_row_stride);
    else
      hipLaunchKernelGGL((scaled_fp8_quant_kernel<scalar_t, false>), grid, block, 0, s, o, in,
                         scale, hidden_size, in_row_stride, out_row_stride);
    return check_launch("static_scaled_fp8_quant");
  });
}

int mi355x_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale,
                                    int num_tokens, int hidden_size,
                                    int64_t in_row_stride, int64_t out_row_stride,
                                    int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "dynamic_scaled_fp8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scale, MI355X_EINVAL, "dynamic_scaled_fp8_quant: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    const scalar_t* in = static_cast<const scalar_t*>(input);
    uint8_t* o = static_cast<uint8_t*>(out);
    const bool vec = row_vec_ok<scalar_t>(in, o, hidden_size, in_row_stride, out_row_stride);
    dim3 grid(num_tokens), block(row_threads(hidden_size, vec ? 16 / sizeof(scalar_t) : 1));
    if (vec) {
      hipLaunchKernelGGL((absmax_to_scale_kernel<scalar_t, true>), grid, block, 0, s, scale, in,
                         hidden_size, in_row_stride);
      hipLaunchKernelGGL((scaled_fp8_quant_kernel<scalar_t, true>), grid, block, 0, s, o, in,
                         scale, hidden_size, in_row_stride, out_row_stride);
    } else {
      hipLaunchKernelGGL((absmax_to_scale_kernel<scalar_t, false>), grid, block, 0, s, scale,
                         in, hidden_size, in_row_stride);
      hipLaunchKernelGGL((scaled_fp8_quant_kernel<scalar_t, false>), grid, block, 0, s, o, in,
                         scale, hidden_size, in_row_stride, out_row_stride);
    }
    return check_launch("dynamic_scaled_fp8_quant");
  });
}

int mi355x_dynamic_per_token_scaled_fp8_quant(void* out, const void* input,
                                              float* scales, const float* scale_ub,
                                              int num_tokens, int hidden_size,
                                              int64_t in_row_stride,
                                              int64_t out_row_stride, int dtype,
                                              mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "dynamic_per_token_scaled_fp8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scales, MI355X_EINVAL,
                 "dynamic_per_token_scaled_fp8_quant: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    const scalar_t* in = static_cast<const scalar_t*>(input);
    uint8_t* o = static_cast<uint8_t*>(out);
    const bool vec = row_vec_ok<scalar_t>(in, o, hidden_size, in_row_stride, out_row_stride);
    dim3 grid(num_tokens), block(row_threads(hidden_size, vec ? 16 / sizeof(scalar_t) : 1));
    if (vec)
      hipLaunchKernelGGL((per_token_fp8_quant_kernel<scalar_t, true>), grid, block, 0, s, o,
                         scales, in, scale_ub, hidden_size, in_row_stride, out_row_stride);
    else
      hipLaunchKernelGGL((per_token_fp8_quant_kernel<scalar_t, false>), grid, block, 0, s, o,
                         scales, in, scale_ub, hidden_size, in_row_stride, out_row_stride);
    return check_launch("dynamic_per_token_scaled_fp8_quant");
  });
}

}  // extern "C"
