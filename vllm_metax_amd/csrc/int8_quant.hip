// int8_quant.hip — activation -> int8 quantisation (symmetric) for the W8A8 path.
//
// Reference semantics restated (csrc/quantization/compressed_tensors/int8_quant_kernels.cu:12-22,
// :50-68 static, :94-135 dynamic per token):
//   rn(x)   = clamp(round-half-even(x), -127, 127)          (this fork clamps at -127, not -128)
//   static : q = rn( float(x) / scale )
//   dynamic: absmax over the token row;  scale_out[token] = absmax / 127;
//            q = rn( float(x) * (absmax == 0 ? 0 : 127 / absmax) )
// The asymmetric (azp) variants (:70-92, :137-…) are not implemented: callers get an error.
// One workgroup per token row, 16-B loads / 8-B stores when the row is aligned.
#include "common.cuh"

namespace mi355x {

__device__ __forceinline__ int8_t float_to_int8_rn(float x) {
  int v = (int)__builtin_rintf(x);      // v_rndne_f32 + v_cvt_i32_f32: round half to even
  v = v < 127 ? v : 127;
  v = v > -127 ? v : -127;
  return (int8_t)v;
}

template <typename T, bool VEC, bool DIVIDE>
__device__ __forceinline__ void int8_quant_row(int8_t* out_row, const T* row, int n, float s) {
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    for (int i = threadIdx.x * V; i < n; i += blockDim.x * V) {
      T v[V];
      int8_t q[V];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(row + i);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float x = to_f32(v[j]);
        q[j] = float_to_int8_rn(DIVIDE ? x / s : x * s);
      }
      if constexpr (V == 8) {
        *reinterpret_cast<uint2*>(out_row + i) = *reinterpret_cast<const uint2*>(q);
      } else {
        *reinterpret_cast<uint32_t*>(out_row + i) = *reinterpret_cast<const uint32_t*>(q);
      }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float x = to_f32(row[i]);
      out_row[i] = float_to_int8_rn(DIVIDE ? x / s : x * s);
    }
  }
}

template <typename T, bool VEC>
__global__ void static_int8_quant_kernel(int8_t* __restrict__ out, const T* __restrict__ in,
                                         const float* __restrict__ scale, int hidden,
                                         int64_t in_stride) {
  const int64_t row = blockIdx.x;
  int8_quant_row<T, VEC, true>(out + row * hidden, in + row * in_stride, hidden, *scale);
}

template <typename T, bool VEC>
__global__ void dynamic_int8_quant_kernel(int8_t* __restrict__ out, const T* __restrict__ in,
                                          float* __restrict__ scales, int hidden,
                                          int64_t in_stride) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const T* r = in + row * in_stride;
  float m = 0.f;
  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    for (int i = threadIdx.x * V; i < hidden; i += blockDim.x * V) {
      T v[V];
      *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(r + i);
#pragma unroll
      for (int j = 0; j < V; ++j) m = fmaxf(m, fabsf(to_f32(v[j])));
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) m = fmaxf(m, fabsf(to_f32(r[i])));
  }
  const float absmax = block_reduce<true>(m, red);
  if (threadIdx.x == 0) scales[row] = absmax / 127.f;
  const float inv_s = absmax == 0.f ? 0.f : 127.f / absmax;
  int8_quant_row<T, VEC, false>(out + row * hidden, r, hidden, inv_s);
}

}  // namespace mi355x

using namespace mi355x;

static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int mi355x_static_scaled_int8_quant(void* out, const void* input, const float* scale,
                                               int num_tokens, int hidden_size,
                                               int64_t input_stride, int dtype,
                                               mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL, "static_scaled_int8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scale, MI355X_EINVAL, "static_scaled_int8_quant: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&]() -> int {
    constexpr int V = 16 / sizeof(scalar_t);
    const scalar_t* in = static_cast<const scalar_t*>(input);
    const bool vec = hidden_size % V == 0 && input_stride % V == 0 && al16p(in) &&
                     (reinterpret_cast<uintptr_t>(out) % V) == 0;
    const int threads = hidden_size / (vec ? V : 1) >= 256 ? 256 : 64;
    if (vec) {
      hipLaunchKernelGGL((static_int8_quant_kernel<scalar_t, true>), dim3(num_tokens), dim3(threads), 0, s,
                         static_cast<int8_t*>(out), in, scale, hidden_size, input_stride);
    } else {
      hipLaunchKernelGGL((static_int8_quant_kernel<scalar_t, false>), dim3(num_tokens), dim3(threads), 0, s,
                         static_cast<int8_t*>(out), in, scale, hidden_size, input_stride);
    }
    return check_launch("static_scaled_int8_quant");
  });
}

extern "C" int mi355x_dynamic_scaled_int8_quant(void* out, const void* input, float* scales,
                                                int num_tokens, int hidden_size,
                                                int64_t input_stride, int dtype,
                                                mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL, "dynamic_scaled_int8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && scales, MI355X_EINVAL, "dynamic_scaled_int8_quant: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&]() -> int {
    constexpr int V = 16 / sizeof(scalar_t);
    const scalar_t* in = static_cast<const scalar_t*>(input);
    const bool vec = hidden_size % V == 0 && input_stride % V == 0 && al16p(in) &&
                     (reinterpret_cast<uintptr_t>(out) % V) == 0;
    const int threads = hidden_size / (vec ? V : 1) >= 256 ? 256 : 64;
    if (vec) {
      hipLaunchKernelGGL((dynamic_int8_quant_kernel<scalar_t, true>), dim3(num_tokens), dim3(threads), 0, s,
                         static_cast<int8_t*>(out), in, scales, hidden_size, input_stride);
    } else {
      hipLaunchKernelGGL((dynamic_int8_quant_kernel<scalar_t, false>), dim3(num_tokens), dim3(threads), 0, s,
                         static_cast<int8_t*>(out), in, scales, hidden_size, input_stride);
    }
    return check_launch("dynamic_scaled_int8_quant");
  });
}
