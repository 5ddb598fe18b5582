// layernorm.hip — RMSNorm family for gfx950: one workgroup per token row, the row
// is read from HBM exactly once (16 B per lane) and kept in registers as fp32
// between the variance pass and the normalise/quantise pass.
//
// Rounding points follow the reference kernels exactly:
//   rms_norm / fused_add_rms_norm      csrc/layernorm_kernels.cu:12-41, 47-137
//     z = T(in + res) (fused only, rounded to T);  x = float(z)
//     out = T( float(T(x * rsqrt(mean(x^2)+eps))) * float(w) )
//   *_static_fp8_quant                 csrc/layernorm_quant_kernels.cu:20-164
//     fp8 = sat_e4m3( float(out) * (1/scale) )
//   rms_norm_dynamic_per_token_quant   csrc/quantization/fused_kernels/
//     layernorm_utils.cuh:17-115 — x = float(in) + float(res) is NOT rounded before
//     the norm; residual = T(x); scale = max(min(absmax,ub)/448, 1/(448*512));
//     fp8 = sat_e4m3( float(out) / scale )   (true division).
#include <cstdlib>

#include "common.cuh"

namespace mi355x {

enum NormOut { kOutT = 0, kOutFp8Static = 1, kOutFp8Dynamic = 2 };

// SLABS: the input row is the split-K partial slabs of the preceding GEMM (decode).  A template switch, not a run-time
// branch: the slab loads keep up to 8 x V values in flight per chunk (64 VGPRs), and as a run-time branch they set the
// register allocation of the prefill-sized launches too (MAXC = 2: 101 VGPRs = two 512-thread workgroups per CU).
template <typename T, int V, int MAXC, bool FUSED_ADD, int OUT, bool SLABS = false>
__global__ void rms_norm_kernel(void* __restrict__ out_v,  // T* or uint8_t*
                                T* __restrict__ input, int64_t input_stride,
                                T* __restrict__ residual, const T* __restrict__ weight,
                                const float* __restrict__ scale_in,   // static scale
                                float* __restrict__ scales_out,       // dynamic scales
                                const float* __restrict__ scale_ub, float epsilon,
                                int hidden_size,
                                // optional: the input row is T(slab[0] + .. + slab[sk-1]) (fp32
                                // split-K partials of the preceding GEMM, see mi355x_*_gemm_deferred)
                                const float* __restrict__ slabs, int sk, int64_t slab_stride,
                                SlabScales slab_scales = SlabScales{nullptr, nullptr, 0, 0}) {
  __shared__ float red[16];
  __shared__ float s_bcast;
  const int64_t row = blockIdx.x;
  T* in_row = input + row * input_stride;
  T* res_row = residual ? residual + row * hidden_size : nullptr;
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;

  float x[MAXC][V];
  float ss = 0.f;
  // the weights are fetched with the row (not after the reduction: a decode-sized launch is a chain of
  // memory round trips, and this one would be the second)
  T wreg[MAXC][V];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int idx = (c * nthreads + tid) * V;
    if (idx < hidden_size) {
      if constexpr (V > 1) *reinterpret_cast<uint4*>(wreg[c]) = *reinterpret_cast<const uint4*>(weight + idx);
      else wreg[c][0] = weight[idx];
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int idx = (c * nthreads + tid) * V;
    if (idx < hidden_size) {
      T iv[V];
      T rv[V];
      if constexpr (SLABS) {
        // same arithmetic as w4a16_sum_slabs_kernel: fp32 sum in slab order, one rounding to T (fp8 GEMM slabs:
        // times the GEMM's scales first = its finish kernel, common.cuh)
        slab_values<T, V>(slabs + row * hidden_size + idx, sk, slab_stride, slab_scales, row, idx, iv);
        if (res_row) {
          if constexpr (V > 1) *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(res_row + idx);
          else rv[0] = res_row[idx];
        }
      } else if constexpr (V > 1) {
        *reinterpret_cast<uint4*>(iv) = *reinterpret_cast<const uint4*>(in_row + idx);
        if (res_row) *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(res_row + idx);
      } else {
        iv[0] = in_row[idx];
        if (res_row) rv[0] = res_row[idx];
      }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float v = to_f32(iv[j]);
        if (res_row) {
          v += to_f32(rv[j]);
          const T z = from_f32<T>(v);
          rv[j] = z;
          if constexpr (FUSED_ADD) v = to_f32(z);  // norm of the ROUNDED sum
        }
        x[c][j] = v;
        ss += v * v;
      }
      if (res_row) {
        if constexpr (V > 1) {
          *reinterpret_cast<uint4*>(res_row + idx) = *reinterpret_cast<const uint4*>(rv);
        } else {
          res_row[idx] = rv[0];
        }
      }
    }
  }
  ss = block_reduce<false>(ss, red);
  const float inv_rms = rsqrtf(ss / hidden_size + epsilon);

  // normalise (T rounding before and after the weight multiply)
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int idx = (c * nthreads + tid) * V;
    if (idx < hidden_size) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const T n = from_f32_rounded<T>(x[c][j] * inv_rms);
        const T o = mul_t<T>(n, wreg[c][j]);
        x[c][j] = to_f32(o);
        amax = fmaxf(amax, fabsf(x[c][j]));
      }
    }
  }

  float q_mul = 1.f;   // multiply (static) ...
  float q_div = 1.f;   // ... or divide (dynamic)
  if constexpr (OUT == kOutFp8Static) {
    q_mul = 1.0f / *scale_in;
  } else if constexpr (OUT == kOutFp8Dynamic) {
    amax = block_reduce<true>(amax, red);
    if (tid == 0) {
      float s = scale_ub ? fminf(amax, *scale_ub) : amax;
      s = fmaxf(s / kFp8Max, kFp8MinScale);
      scales_out[row] = s;
      s_bcast = s;
    }
    __syncthreads();
    q_div = s_bcast;
  }
  const RowDiv rdiv = make_row_div(q_div);   // (the true division, three operations per element: common.cuh)

#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int idx = (c * nthreads + tid) * V;
    if (idx < hidden_size) {
      if constexpr (OUT == kOutT) {
        // fused_add writes back into `input` (strided); rms_norm into `out` (dense)
        T* dst = FUSED_ADD ? (in_row + idx)
                           : (static_cast<T*>(out_v) + row * hidden_size + idx);
        T ov[V];
#pragma unroll
        for (int j = 0; j < V; ++j) ov[j] = from_f32<T>(x[c][j]);  // exact: already T
        if constexpr (V > 1) {
          *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(ov);
        } else {
          dst[0] = ov[0];
        }
      } else {
        uint8_t* dst = static_cast<uint8_t*>(out_v) + row * hidden_size + idx;
        float y[V];
#pragma unroll
        for (int j = 0; j < V; ++j) y[j] = (OUT == kOutFp8Static) ? x[c][j] * q_mul : row_div(x[c][j], rdiv);
        if constexpr (V == 8) {
          *reinterpret_cast<uint2*>(dst) = f32x8_to_fp8x8_sat(y);
        } else if constexpr (V == 4) {
          *reinterpret_cast<uint32_t*>(dst) = f32x4_to_fp8x4_sat(y);
        } else {
          dst[0] = f32_to_fp8_sat(y[0]);
        }
      }
    }
  }
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, bool FUSED_ADD, int OUT>
static int launch_norm(void* out, T* input, int64_t input_stride, T* residual,
                       const T* weight, const float* scale_in, float* scales_out,
                       const float* scale_ub, float eps, int num_tokens, int hidden,
                       hipStream_t s, const char* name, const float* slabs = nullptr, int sk = 0,
                       int64_t slab_stride = 0, SlabScales slab_scales = SlabScales{nullptr, nullptr, 0, 0}) {
  constexpr int V = 16 / sizeof(T);
  const bool vec = (hidden % V == 0) && (input_stride % V == 0) && al16(input) &&
                   al16(weight) && (!residual || al16(residual)) &&
                   (OUT != kOutT || FUSED_ADD || al16(out)) &&
                   (OUT == kOutT || (reinterpret_cast<uintptr_t>(out) % V == 0));
  const int units = vec ? hidden / V : hidden;
  int threads = num_tokens < 256 ? 1024 : 256;
  const int need = ((units + 63) / 64) * 64;
  if (threads > need) threads = need;
  auto chunks = [&](int t) { return (units + t - 1) / t; };
  const int maxc = vec ? 4 : 32;
  // four 16-byte chunks per thread (hidden 8192 at 256 threads) need 128 VGPRs + 9-10 spilled: two chunks at twice
  // the threads fit (101) — Llama-3-70B / Qwen2-72B prefill norms ran at 38 % of the HBM roofline with the spills
  // (profiles/r03_rank_of_8_70b_fp8.json)
  static const int kVecChunks = [] { const char* e = getenv("MI355X_NORM_CHUNKS"); return e ? atoi(e) : 2; }();
  while (chunks(threads) > (vec ? kVecChunks : maxc) && threads < 1024) threads *= 2;
  if (threads > 1024) threads = 1024;
  MI355X_REQUIRE(chunks(threads) <= maxc, MI355X_EUNSUPPORTED,
                 "%s: hidden_size %d too large (max %d)", name, hidden, maxc * 1024 * (vec ? V : 1));
  const int c = chunks(threads);
  dim3 grid(num_tokens), block(threads);
#define LAUNCH_NORM(VV, CC)                                                                        \
  do {                                                                                             \
    if (slabs != nullptr)                                                                          \
      hipLaunchKernelGGL((rms_norm_kernel<T, VV, CC, FUSED_ADD, OUT, true>), grid, block, 0, s, out, \
                         input, input_stride, residual, weight, scale_in, scales_out,               \
                         scale_ub, eps, hidden, slabs, sk, slab_stride, slab_scales);               \
    else                                                                                           \
      hipLaunchKernelGGL((rms_norm_kernel<T, VV, CC, FUSED_ADD, OUT, false>), grid, block, 0, s, out, \
                         input, input_stride, residual, weight, scale_in, scales_out,               \
                         scale_ub, eps, hidden, slabs, sk, slab_stride, slab_scales);               \
  } while (0)
  if (vec) {
    if (c <= 1) LAUNCH_NORM(V, 1);
    else if (c <= 2) LAUNCH_NORM(V, 2);
    else LAUNCH_NORM(V, 4);
  } else {
    if (c <= 4) LAUNCH_NORM(1, 4);
    else LAUNCH_NORM(1, 32);
  }
#undef LAUNCH_NORM
  return check_launch(name);
}

// ---- rms_norm / fused_add_rms_norm whose output is the prefill GEMM's operand image ----------------
// (mi355x_rms_norm_image: saves the activation re-tiling launch, pack_a_kernel, in front of the qkv and
// gate_up GEMMs of a prefill chunk.)  One workgroup = one 16-row tile of the image, one wave per row, so the
// 16-byte slots a workgroup writes complete whole 1-KiB pieces inside one XCD's L2.
// Bit-identical to rms_norm_kernel<T, 8, MAXC, ..> at 256 threads (what launch_norm picks for >= 256 tokens):
// lane l replays the partial sums of that kernel's threads l, l + 64, l + 128, l + 192 (chunks t + 256 c in
// the same order), the four wave sums use the same butterfly, and block_reduce's lane tree adds them as
// (W0 + W2) + (W1 + W3).
__device__ __forceinline__ int norm_frag_swz(int lr, int lc) {   // = frag_swz of w4a16.cuh (operand image slot)
  return lr * 16 + (lc ^ (((lr & 1) * 12) | (lr & 2)));
}
template <typename T, int MAXC, bool FUSED_ADD>
__global__ __launch_bounds__(1024) void rms_norm_image_kernel(
    T* __restrict__ image, const T* __restrict__ input, int64_t input_stride, T* __restrict__ residual,
    const T* __restrict__ weight, float epsilon, int num_tokens, int hidden_size) {
  constexpr int V = 8;
  const int lane = threadIdx.x & 63;
  const int lc = threadIdx.x >> 6;                      // row of the tile = wave
  const int64_t row = (int64_t)blockIdx.x * 16 + lc;
  const bool live = row < num_tokens;
  const T* in_row = input + row * input_stride;
  T* res_row = residual + row * hidden_size;
  float x[4][MAXC][V];
  T wreg[4][MAXC][V];
  float part[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int idx = (c * 256 + 64 * k + lane) * V;
      T iv[V], rv[V];
      if (live) {
        *reinterpret_cast<uint4*>(iv) = *reinterpret_cast<const uint4*>(in_row + idx);
        if constexpr (FUSED_ADD) *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(res_row + idx);
      } else {
        *reinterpret_cast<uint4*>(iv) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(rv) = make_uint4(0, 0, 0, 0);
      }
      *reinterpret_cast<uint4*>(wreg[k][c]) = *reinterpret_cast<const uint4*>(weight + idx);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float v = to_f32(iv[j]);
        if constexpr (FUSED_ADD) {
          v += to_f32(rv[j]);
          const T z = from_f32<T>(v);
          rv[j] = z;
          v = to_f32(z);  // norm of the ROUNDED sum
        }
        x[k][c][j] = v;
        ss += v * v;
      }
      if constexpr (FUSED_ADD) {
        if (live) *reinterpret_cast<uint4*>(res_row + idx) = *reinterpret_cast<const uint4*>(rv);
      }
    }
    part[k] = wave_sum(ss);
  }
  const float ss = (part[0] + part[2]) + (part[1] + part[3]);
  const float inv_rms = rsqrtf(ss / hidden_size + epsilon);
  const int kt32 = hidden_size >> 5;
  uint4* img = reinterpret_cast<uint4*>(image) + (int64_t)blockIdx.x * kt32 * 64;
  // One (k, c) step of the 16 waves = chunks 256 c + 64 k .. + 63 of 16 rows = 16 COMPLETE pieces (k tiles
  // 64 c + 16 k .. + 15): exchanged through LDS so that every wave stores one whole 1-KiB piece linearly
  // (direct 16-byte stores at slot addresses ran the kernel at 3.3 TB/s instead of the row-major 5.7).
  __shared__ uint4 stage[2][16 * 64];
  int buf = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      T ov[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const T n = from_f32_rounded<T>(x[k][c][j] * inv_rms);
        ov[j] = live ? mul_t<T>(n, wreg[k][c][j]) : from_f32<T>(0.f);   // rows >= num_tokens of the tile: zero
      }
      // chunk 256 c + 64 k + lane of row lc: piece (lane >> 2) of this step, k group lane & 3
      stage[buf][(lane >> 2) * 64 + norm_frag_swz(lane & 3, lc)] = *reinterpret_cast<const uint4*>(ov);
      __syncthreads();   // (two buffers: the next step writes the other one; this one is rewritten two steps on,
                         //  behind the next step's barrier, which every wave reaches after its reads below)
      img[(64 * c + 16 * k + lc) * 64 + lane] = stage[buf][lc * 64 + lane];
      buf ^= 1;
    }
  }
}

// returns 1 when the image form does not apply (caller: rms_norm / fused_add_rms_norm + the GEMM's own re-tiling)
template <typename T, bool FUSED_ADD>
static int launch_norm_image(T* image, const T* input, int64_t input_stride, T* residual, const T* weight,
                             float eps, int num_tokens, int hidden, hipStream_t s, const char* name) {
  if (num_tokens < 256 || (hidden != 2048 && hidden != 4096) || input_stride % 8 != 0 || !al16(input) ||
      !al16(weight) || !al16(image) || (FUSED_ADD && !al16(residual)))
    return 1;
  dim3 grid((num_tokens + 15) / 16), block(1024);
  if (hidden == 2048) {
    hipLaunchKernelGGL((rms_norm_image_kernel<T, 1, FUSED_ADD>), grid, block, 0, s, image, input, input_stride,
                       residual, weight, eps, num_tokens, hidden);
  } else {
    hipLaunchKernelGGL((rms_norm_image_kernel<T, 2, FUSED_ADD>), grid, block, 0, s, image, input, input_stride,
                       residual, weight, eps, num_tokens, hidden);
  }
  return check_launch(name);
}

}  // namespace mi355x

using namespace mi355x;

extern "C" {

int mi355x_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                    int num_tokens, int hidden_size, int64_t input_stride, int dtype,
                    mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL, "rms_norm: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && weight, MI355X_EINVAL, "rms_norm: null pointer");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    return launch_norm<scalar_t, false, kOutT>(
        out, const_cast<scalar_t*>(static_cast<const scalar_t*>(input)), input_stride, nullptr,
        static_cast<const scalar_t*>(weight), nullptr, nullptr, nullptr, epsilon, num_tokens,
        hidden_size, static_cast<hipStream_t>(stream), "rms_norm");
  });
}

int mi355x_fused_add_rms_norm(void* input, void* residual, const void* weight,
                              float epsilon, int num_tokens, int hidden_size,
                              int64_t input_stride, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "fused_add_rms_norm: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(input && residual && weight, MI355X_EINVAL,
                 "fused_add_rms_norm: null pointer");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    return launch_norm<scalar_t, true, kOutT>(
        nullptr, static_cast<scalar_t*>(input), input_stride, static_cast<scalar_t*>(residual),
        static_cast<const scalar_t*>(weight), nullptr, nullptr, nullptr, epsilon, num_tokens,
        hidden_size, static_cast<hipStream_t>(stream), "fused_add_rms_norm");
  });
}

int mi355x_fused_add_rms_norm_slabs(void* input, void* residual, const void* weight,
                                    const float* slabs, int sk, float epsilon, int num_tokens,
                                    int hidden_size, int64_t input_stride, int dtype,
                                    mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0 && sk >= 0, MI355X_EINVAL,
                 "fused_add_rms_norm_slabs: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(input && residual && weight, MI355X_EINVAL,
                 "fused_add_rms_norm_slabs: null pointer");
  MI355X_REQUIRE(sk == 0 || (slabs && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 &&
                             hidden_size % 8 == 0),
                 MI355X_EINVAL, "fused_add_rms_norm_slabs: slabs must be 16-byte aligned, hidden %% 8 == 0");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    return launch_norm<scalar_t, true, kOutT>(
        nullptr, static_cast<scalar_t*>(input), input_stride, static_cast<scalar_t*>(residual),
        static_cast<const scalar_t*>(weight), nullptr, nullptr, nullptr, epsilon, num_tokens,
        hidden_size, static_cast<hipStream_t>(stream), "fused_add_rms_norm_slabs",
        sk > 0 ? slabs : nullptr, sk, (int64_t)num_tokens * hidden_size);
  });
}

int mi355x_rms_norm_image(void* image, const void* input, const void* weight, float epsilon, int num_tokens,
                          int hidden_size, int64_t input_stride, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL, "rms_norm_image: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(image && input && weight, MI355X_EINVAL, "rms_norm_image: null pointer");
  if (dtype != MI355X_BF16 && dtype != MI355X_F16) return 1;
  return MI355X_DISPATCH_HALF(dtype, [&] {
    return launch_norm_image<scalar_t, false>(static_cast<scalar_t*>(image), static_cast<const scalar_t*>(input),
                                              input_stride, nullptr, static_cast<const scalar_t*>(weight), epsilon,
                                              num_tokens, hidden_size, static_cast<hipStream_t>(stream),
                                              "rms_norm_image");
  });
}

int mi355x_fused_add_rms_norm_image(void* image, const void* input, void* residual, const void* weight,
                                    float epsilon, int num_tokens, int hidden_size, int64_t input_stride,
                                    int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL, "fused_add_rms_norm_image: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(image && input && residual && weight, MI355X_EINVAL, "fused_add_rms_norm_image: null pointer");
  if (dtype != MI355X_BF16 && dtype != MI355X_F16) return 1;
  return MI355X_DISPATCH_HALF(dtype, [&] {
    return launch_norm_image<scalar_t, true>(static_cast<scalar_t*>(image), static_cast<const scalar_t*>(input),
                                             input_stride, static_cast<scalar_t*>(residual),
                                             static_cast<const scalar_t*>(weight), epsilon, num_tokens, hidden_size,
                                             static_cast<hipStream_t>(stream), "fused_add_rms_norm_image");
  });
}

int mi355x_rms_norm_static_fp8_quant(void* out, const void* input, const void* weight,
                                     const float* scale, float epsilon, int num_tokens,
                                     int hidden_size, int64_t input_stride, int dtype,
                                     mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "rms_norm_static_fp8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && weight && scale, MI355X_EINVAL,
                 "rms_norm_static_fp8_quant: null pointer");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    return launch_norm<scalar_t, false, kOutFp8Static>(
        out, const_cast<scalar_t*>(static_cast<const scalar_t*>(input)), input_stride, nullptr,
        static_cast<const scalar_t*>(weight), scale, nullptr, nullptr, epsilon, num_tokens,
        hidden_size, static_cast<hipStream_t>(stream), "rms_norm_static_fp8_quant");
  });
}

int mi355x_fused_add_rms_norm_static_fp8_quant(void* out, void* input, void* residual,
                                               const void* weight, const float* scale,
                                               float epsilon, int num_tokens,
                                               int hidden_size, int64_t input_stride,
                                               int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "fused_add_rms_norm_static_fp8_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && residual && weight && scale, MI355X_EINVAL,
                 "fused_add_rms_norm_static_fp8_quant: null pointer");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    return launch_norm<scalar_t, true, kOutFp8Static>(
        out, static_cast<scalar_t*>(input), input_stride, static_cast<scalar_t*>(residual),
        static_cast<const scalar_t*>(weight), scale, nullptr, nullptr, epsilon, num_tokens,
        hidden_size, static_cast<hipStream_t>(stream), "fused_add_rms_norm_static_fp8_quant");
  });
}

int mi355x_rms_norm_dynamic_per_token_quant(void* out, const void* input,
                                            const void* weight, float* scales,
                                            float epsilon, const float* scale_ub,
                                            void* residual, int num_tokens,
                                            int hidden_size, int dtype,
                                            mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0, MI355X_EINVAL,
                 "rms_norm_dynamic_per_token_quant: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && input && weight && scales, MI355X_EINVAL,
                 "rms_norm_dynamic_per_token_quant: null pointer");
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    // FUSED_ADD=false: the norm uses the UNROUNDED fp32 sum (layernorm_utils.cuh:101-107)
    return launch_norm<scalar_t, false, kOutFp8Dynamic>(
        out, const_cast<scalar_t*>(static_cast<const scalar_t*>(input)), hidden_size,
        static_cast<scalar_t*>(residual), static_cast<const scalar_t*>(weight), nullptr, scales,
        scale_ub, epsilon, num_tokens, hidden_size, static_cast<hipStream_t>(stream),
        "rms_norm_dynamic_per_token_quant");
  });
}

/* rms_norm_dynamic_per_token_quant whose input rows are still the split-K partial slabs of an fp8 GEMM
 * (mi355x_scaled_mm_fp8_deferred): input = T(sum of slabs * a_scale[token] * b_scale[column]), then the op above —
 * bit-identical to the GEMM's finish launch followed by it.  `residual` as above (updated in place). */
int mi355x_rms_norm_dynamic_per_token_quant_slabs(void* out, const float* slabs, int sk,
                                                  const float* a_scales, int a_scales_numel,
                                                  const float* b_scales, int b_scales_numel,
                                                  const void* weight, float* scales, float epsilon,
                                                  const float* scale_ub, void* residual, int num_tokens,
                                                  int hidden_size, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && hidden_size > 0 && sk > 0, MI355X_EINVAL,
                 "rms_norm_dynamic_per_token_quant_slabs: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(out && slabs && weight && scales && a_scales && b_scales, MI355X_EINVAL,
                 "rms_norm_dynamic_per_token_quant_slabs: null pointer");
  MI355X_REQUIRE((reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && hidden_size % 8 == 0, MI355X_EINVAL,
                 "rms_norm_dynamic_per_token_quant_slabs: slabs must be 16-byte aligned, hidden %% 8 == 0");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == num_tokens) &&
                     (b_scales_numel == 1 || b_scales_numel == hidden_size),
                 MI355X_EINVAL, "rms_norm_dynamic_per_token_quant_slabs: scales per-tensor or per-token / per-column");
  return MI355X_DISPATCH_HALF(dtype, [&] {
    // `input` is only an address for the vector-path checks here: every row comes from the slabs
    return launch_norm<scalar_t, false, kOutFp8Dynamic>(
        out, static_cast<scalar_t*>(const_cast<void*>(static_cast<const void*>(slabs))), hidden_size,
        static_cast<scalar_t*>(residual), static_cast<const scalar_t*>(weight), nullptr, scales,
        scale_ub, epsilon, num_tokens, hidden_size, static_cast<hipStream_t>(stream),
        "rms_norm_dynamic_per_token_quant_slabs", slabs, sk, (int64_t)num_tokens * hidden_size,
        SlabScales{a_scales, b_scales, a_scales_numel > 1, b_scales_numel > 1});
  });
}

}  // extern "C"
