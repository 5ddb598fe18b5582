// common.cuh — wave64 / gfx950 helpers shared by the hot-path kernels.
// Written for CDNA4 only: 64-lane wavefronts, DPP/permlane cross-lane ops,
// native __bf16 / _Float16 conversions (v_cvt_pk_bf16_f32 on gfx950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355x_hotpath.h"

namespace mi355x {

constexpr int kWave = 64;

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define MI355X_REQUIRE(cond, code, ...)  \
  do {                                   \
    if (!(cond)) {                       \
      ::mi355x::set_error(__VA_ARGS__);  \
      return (code);                     \
    }                                    \
  } while (0)

// One-time per-DEVICE setup of a kernel (hipFuncSetAttribute is a per-device property): a bit per
// device ordinal.  The header promises "no global state": this is a cache of a fact about the device,
// not state of a call; setting the attribute twice (two threads racing) is harmless.
struct PerDeviceOnce {
  unsigned long long done = 0;
  bool need(int* dev) {
    *dev = 0;
    (void)hipGetDevice(dev);
    return ((done >> (*dev & 63)) & 1ull) == 0;
  }
  void mark(int dev) { done |= 1ull << (dev & 63); }
};

// ---- asynchronous global -> LDS copies (LDS-DMA) -----------------------------------------
// One wave copies 64 x 16 B: lane l reads 16 B at `gptr` (per lane) and the hardware writes
// them at LDS byte address lds_base + 16 * l (lds_base is wave-uniform, in M0).
// Spelled as inline asm on purpose: with the builtin the compiler's wait-count pass cannot
// tell which LDS bytes a pending DMA will write and puts `s_waitcnt vmcnt(0)` in front of the
// next ds_read of ANY LDS address, which serialises every multi-stage pipeline.  These copies
// are invisible to that pass, so the kernel must order them itself: lds_dma_wait<N>() (at most
// N of this wave's copies still in flight; they complete in issue order) and then a barrier
// before another wave reads the bytes.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_dma16(const void* gptr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(gptr), "s"(lds_base)
               : "memory");  // (M0 is reserved: the compiler never keeps a value in it)
}
// same copy, source = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: no 64-bit
// VGPR address and no 64-bit VALU add per copy (the base advances on the scalar unit)
__device__ __forceinline__ void lds_dma16_s(const void* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_base)
               : "memory");
}
__device__ __forceinline__ void lds_dma16_s_nt(const void* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
               :
               : "v"(voff), "s"(sbase), "s"(lds_base)
               : "memory");
}
template <int N>
__device__ __forceinline__ void lds_dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// ---- scalar conversions ------------------------------------------------------
template <typename T>
__device__ __forceinline__ float to_f32(T v) {
  return static_cast<float>(v);
}
// from_f32 of a value the compiler must MATERIALISE in fp32 first: for T = f16, hipcc folds
// T(float(a_f16) * b_f32) into v_fma_mixlo_f16, which rounds the exact product ONCE to f16, while the reference
// (fp32 multiply, then the scalar_t cast) rounds twice — about one output in 8000 differs by an ulp.
template <typename T>
__device__ __forceinline__ T from_f32_rounded(float v) {
  asm volatile("" : "+v"(v));
  return static_cast<T>(v);
}
// Round-to-nearest-even conversions (c10::BFloat16 / c10::Half semantics).
template <typename T>
__device__ __forceinline__ T from_f32(float v) {
  return static_cast<T>(v);
}

// bf16 pair <-> float without going through the type system (used on packed loads)
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) {
  return __uint_as_float(packed << 16);
}
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) {
  return __uint_as_float(packed & 0xffff0000u);
}

// `a op b` evaluated the way c10 scalar types do it: in float, then rounded to T.
// Contraction is switched off here: hipcc otherwise narrows these to native f16 ops and
// then fuses T(a*b) - T(c*d) into one fma, dropping the reference's intermediate rounding.
template <typename T>
__device__ __forceinline__ T mul_t(T a, T b) {
#pragma clang fp contract(off)
  return from_f32<T>(to_f32(a) * to_f32(b));
}
template <typename T>
__device__ __forceinline__ T add_t(T a, T b) {
#pragma clang fp contract(off)
  return from_f32<T>(to_f32(a) + to_f32(b));
}
template <typename T>
__device__ __forceinline__ T sub_t(T a, T b) {
#pragma clang fp contract(off)
  return from_f32<T>(to_f32(a) - to_f32(b));
}

// ---- 16-byte vector of T -----------------------------------------------------
template <typename T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  union {
    uint4 u;
    T e[N];
  };
};

template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> v;
  v.u = *reinterpret_cast<const uint4*>(p);
  return v;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<uint4*>(p) = v.u;
}

// SiLU in the reference's arithmetic (csrc/activation_kernels.cu:142-147): fp32, rounded to T
template <typename T>
__device__ __forceinline__ T silu_t(T x) {
  const float xf = to_f32(x);
  return from_f32<T>(xf / (1.0f + expf(-xf)));
}

// rotary pair in scalar_t arithmetic (ref: csrc/pos_encoding_kernels.cu:10-34): a rounding after
// every multiply and after the add / sub, like the c10 scalar operators.
template <typename T>
__device__ __forceinline__ void rot_pair(T& x, T& y, T c, T s) {
  const T xn = sub_t<T>(mul_t<T>(x, c), mul_t<T>(y, s));
  const T yn = add_t<T>(mul_t<T>(y, c), mul_t<T>(x, s));
  x = xn;
  y = yn;
}

// ---- wave64 reductions -------------------------------------------------------
// Butterfly over the lanes whose index differs in the bits of `mask_from..32`.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// Block-wide sum / max for blocks of up to 1024 threads (16 waves).
// `smem` must hold >= 16 floats and is reusable after the call returns.
template <bool IS_MAX>
__device__ __forceinline__ float block_reduce(float v, float* smem) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = IS_MAX ? wave_max(v) : wave_sum(v);
  if (nw == 1) return v;
  __syncthreads();  // protect smem from a previous use
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = IS_MAX ? -3.402823466e+38f : 0.f;
  if (lane < nw) r = smem[lane];
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    float o = __shfl_xor(r, m, 64);
    r = IS_MAX ? fmaxf(r, o) : r + o;
  }
  return __shfl(r, 0, 64);
}

// ---- fp8 e4m3fn (OCP) ----------------------------------------------------------
constexpr float kFp8Max = 448.0f;
constexpr float kFp8MinScale = 1.0f / (448.0f * 512.0f);

// clamp to +-448 then RNE convert (c10::Float8_e4m3fn static_cast semantics for
// in-range values; NaN stays NaN).  ref: csrc/quantization/fp8/common.cuh:25-38.
__device__ __forceinline__ uint8_t f32_to_fp8_sat(float x) {
  float r = fmaxf(-kFp8Max, fminf(x, kFp8Max));
  int packed = __builtin_amdgcn_cvt_pk_fp8_f32(r, r, 0, false);
  return static_cast<uint8_t>(packed & 0xff);
}
__device__ __forceinline__ uint16_t f32x2_to_fp8x2_sat(float a, float b) {
  float ra = fmaxf(-kFp8Max, fminf(a, kFp8Max));
  float rb = fmaxf(-kFp8Max, fminf(b, kFp8Max));
  int packed = __builtin_amdgcn_cvt_pk_fp8_f32(ra, rb, 0, false);
  return static_cast<uint16_t>(packed & 0xffff);
}

// ---- x / s for a whole row with ONE divisor (dynamic per-token quantisation) ---------------------------------
// The reference divides (fp8 = sat(float(x) / scale), csrc/quantization/fp8/common.cuh): an IEEE division is ~10
// VALU operations per element and made the fused norm + quant kernels VALU-bound (profiles/r03 notes).  With the
// correctly rounded reciprocal y = RN(1 / s) — one true division per row — Markstein's correction
//   q0 = RN(x y);  r = x - s q0 (exact in an fma);  q = RN(q0 + r y)
// gives the correctly rounded quotient RN(x / s), i.e. the bits of the division, in 3 operations (IBM J. R&D 34(1),
// 1990, Thm 8; Cornea et al. 2002).  Outside the theorem — a divisor whose significand is all ones, non-finite x —
// the division itself runs (row-uniform / practically never taken branches).  Quotients so small that r underflows
// round to the fp8 zero of the same sign either way.
struct RowDiv {
  float s, y;
  bool fast;
};
__device__ __forceinline__ RowDiv make_row_div(float s) {
  RowDiv d;
  d.s = s;
  d.y = 1.0f / s;
  d.fast = (__float_as_uint(s) & 0x7FFFFFu) != 0x7FFFFFu && s > 0.f && s < 3.0e38f;
  return d;
}
__device__ __forceinline__ float row_div(float x, const RowDiv& d) {
  if (!d.fast) return x / d.s;
  float q0 = x * d.y;
  asm volatile("" : "+v"(q0));                 // (the rounded product, in both uses: no contraction into the fmas)
  const float r = __builtin_fmaf(-d.s, q0, x);
  float q = __builtin_fmaf(r, d.y, q0);
  if (!(fabsf(x) <= 3.0e38f)) q = x / d.s;     // inf / NaN: whatever the division gives
  return q;
}
// 8 values -> 8 saturated e4m3 bytes (two packed converts per dword instead of one convert + mask + shift per byte)
__device__ __forceinline__ uint2 f32x8_to_fp8x8_sat(const float (&v)[8]) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fmaxf(-kFp8Max, fminf(v[j], kFp8Max));
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}
__device__ __forceinline__ uint32_t f32x4_to_fp8x4_sat(const float (&v)[4]) {
  float c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = fmaxf(-kFp8Max, fminf(v[j], kFp8Max));
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
  return (uint32_t)w;
}

// e4m3fn byte -> float (exact; v_cvt_f32_fp8)
__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
}

// ---- fp8 e5m2 (OCP "bf8"): the second 8-bit KV-cache format (kv_cache_dtype "fp8_e5m2", upstream vLLM's name; listed
// by the reference at csrc/attention/dtype_fp8.cuh:9-13 and rejected by its dispatch like e4m3).  Saturating finite
// conversion (max 57344), RNE — v_cvt_pk_bf8_f32.
struct e5m2_t {             // cache element tag: one e5m2 byte
  uint8_t v;
  e5m2_t() = default;
  __host__ __device__ explicit e5m2_t(int x) : v(static_cast<uint8_t>(x)) {}
};
constexpr float kBf8Max = 57344.0f;
__device__ __forceinline__ uint8_t f32_to_bf8_sat(float x) {
  float r = fmaxf(-kBf8Max, fminf(x, kBf8Max));
  int packed = __builtin_amdgcn_cvt_pk_bf8_f32(r, r, 0, false);
  return static_cast<uint8_t>(packed & 0xff);
}
__device__ __forceinline__ uint16_t f32x2_to_bf8x2_sat(float a, float b) {
  float ra = fmaxf(-kBf8Max, fminf(a, kBf8Max));
  float rb = fmaxf(-kBf8Max, fminf(b, kBf8Max));
  int packed = __builtin_amdgcn_cvt_pk_bf8_f32(ra, rb, 0, false);
  return static_cast<uint16_t>(packed & 0xffff);
}
__device__ __forceinline__ float bf8_to_f32(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_bf8((int)b, 0);
}
// the two byte formats behind one switch (E5M2 = false: e4m3fn)
template <bool E5M2>
struct Kv8Fmt {
  static __device__ __forceinline__ uint8_t to8(float x) { return E5M2 ? f32_to_bf8_sat(x) : f32_to_fp8_sat(x); }
  static __device__ __forceinline__ uint16_t to8x2(float a, float b) {
    return E5M2 ? f32x2_to_bf8x2_sat(a, b) : f32x2_to_fp8x2_sat(a, b);
  }
  static __device__ __forceinline__ float from8(uint8_t b) { return E5M2 ? bf8_to_f32(b) : fp8_to_f32(b); }
};

// ---- split-K slab sum ------------------------------------------------------------
// acc[j] = slab[0][j] + slab[1][j] + ... + slab[sk-1][j] in THAT order (the consumers of the
// decode GEMM's fp32 partials must all round the same sum), V consecutive floats per lane.  The
// loads of up to 8 slabs are issued before the first add: a loop of load -> add would pay one
// memory round trip per slab, which is most of the time of the launch-bound decode consumers.
template <int V>
__device__ __forceinline__ void sum_slabs(const float* __restrict__ sp, int sk, int64_t stride,
                                          float (&acc)[V]) {
  constexpr int kMax = 8;
  float v[kMax][V];
#pragma unroll
  for (int s = 0; s < kMax; ++s) {
    if (s < sk) {
      if constexpr (V % 4 == 0) {
#pragma unroll
        for (int j = 0; j < V; j += 4) {
          *reinterpret_cast<float4*>(&v[s][j]) =
              *reinterpret_cast<const float4*>(sp + (int64_t)s * stride + j);
        }
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) v[s][j] = sp[(int64_t)s * stride + j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = v[0][j];
#pragma unroll
  for (int s = 1; s < kMax; ++s) {
    if (s < sk) {
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += v[s][j];
    }
  }
  for (int s = kMax; s < sk; ++s) {
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] += sp[(int64_t)s * stride + j];
  }
}

// ---- split-K slabs of a W8A8 (fp8) GEMM ---------------------------------------------
// A decode-sized scaled GEMM may leave its fp32 partial slabs unreduced (mi355x_scaled_mm_fp8_deferred); its
// consumer then produces the value the GEMM's own finish kernel would have stored:
//   T( (slab 0 + slab 1 + ...) * a_scale[row] * b_scale[col] + 0 )      (fp8_gemm.hip: OpFp8::finish, no bias)
// `b == nullptr`: no scales — the w4a16 slabs (T(sum), mi355x_awq_gemm_deferred).
struct SlabScales {
  const float* a;   // activation scales: [rows] (a_per_row) or one value
  const float* b;   // weight scales: [cols] (b_per_col) or one value; nullptr = unscaled slabs
  int a_per_row, b_per_col;
};
__device__ __forceinline__ float scaled_finish_fp8(float acc, float as, float bs, float bi) {
  float r = acc * as * bs + bi;
  // materialised in fp32 (as the oracle's (acc * a_s * b_s + bias).to(dtype): two roundings): without the barrier
  // hipcc folds the f16 cast of SOME call sites into v_fma_mixlo_f16 (one rounding) — the GEMM's own epilogue /
  // finish kernel and the slab consumers of its deferred form must produce the same bits
  asm volatile("" : "+v"(r));
  return r;
}
// V consecutive columns col .. col + V - 1 of row `row` as T
template <typename T, int V>
__device__ __forceinline__ void slab_values(const float* __restrict__ sp, int sk, int64_t stride, const SlabScales& sc,
                                            int64_t row, int col, T (&out)[V]) {
  float acc[V];
  sum_slabs<V>(sp, sk, stride, acc);
  if (sc.b != nullptr) {
    const float as = sc.a[sc.a_per_row ? row : 0];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = scaled_finish_fp8(acc[j], as, sc.b[sc.b_per_col ? col + j : 0], 0.f);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) out[j] = from_f32<T>(acc[j]);
}

// ---- dtype dispatch ------------------------------------------------------------
#define MI355X_DISPATCH_FLOAT(dtype, ...)                                \
  [&]() -> int {                                                         \
    switch (dtype) {                                                     \
      case MI355X_F16: {                                                 \
        using scalar_t = ::mi355x::f16_t;                                \
        return __VA_ARGS__();                                            \
      }                                                                  \
      case MI355X_BF16: {                                                \
        using scalar_t = ::mi355x::bf16_t;                               \
        return __VA_ARGS__();                                            \
      }                                                                  \
      case MI355X_F32: {                                                 \
        using scalar_t = float;                                          \
        return __VA_ARGS__();                                            \
      }                                                                  \
      default:                                                           \
        ::mi355x::set_error("unsupported dtype id %d", (int)(dtype));    \
        return MI355X_EUNSUPPORTED;                                      \
    }                                                                    \
  }()

#define MI355X_DISPATCH_HALF(dtype, ...)                                     \
  [&]() -> int {                                                             \
    switch (dtype) {                                                         \
      case MI355X_F16: {                                                     \
        using scalar_t = ::mi355x::f16_t;                                    \
        return __VA_ARGS__();                                                \
      }                                                                      \
      case MI355X_BF16: {                                                    \
        using scalar_t = ::mi355x::bf16_t;                                   \
        return __VA_ARGS__();                                                \
      }                                                                      \
      default:                                                               \
        ::mi355x::set_error("dtype id %d: only f16/bf16 supported here",     \
                            (int)(dtype));                                   \
        return MI355X_EUNSUPPORTED;                                          \
    }                                                                        \
  }()

inline int dtype_size(int dtype) { return dtype == MI355X_F32 ? 4 : 2; }

}  // namespace mi355x
