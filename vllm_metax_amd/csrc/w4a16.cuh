// w4a16.cuh — pieces shared by the small-M and large-M w4a16 GEMM kernels.
#pragma once
#include "common.cuh"

namespace mi355x {

// AWQ nibble position of natural column j inside a packed word (inverse of the AWQ
// packing order 0,2,4,6,1,3,5,7); the exllama k-row order uses the same table.
__host__ __device__ constexpr int awq_shift(int j) { return (j >> 1) + 4 * (j & 1); }

template <typename T>
struct Mfma;
template <>
struct Mfma<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    bf16x2_t v = {static_cast<bf16_t>(lo), static_cast<bf16_t>(hi)};
    return __builtin_bit_cast(uint32_t, v);
  }
};
template <>
struct Mfma<f16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f16x2_t v = {static_cast<f16_t>(lo), static_cast<f16_t>(hi)};
    return __builtin_bit_cast(uint32_t, v);
  }
};

// byte -> float conversions: hipcc only pattern-matches v_cvt_f32_ubyte0 and builds the other
// three out of shift / bfe / and (3 VALU ops instead of 1), so they are spelled out.
__device__ __forceinline__ float cvt_ubyte0(uint32_t x) {
  float r;  // (as asm too: written as (float)(x & 0xff) the mask is re-derived from the source word)
  asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float cvt_ubyte1(uint32_t x) {
  float r;
  asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float cvt_ubyte2(uint32_t x) {
  float r;
  asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float cvt_ubyte3(uint32_t x) {
  float r;
  asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Dequantise one shuffled word (8 consecutive k of one column) into an MFMA B fragment:
// value_j = fma(float(q_j), s, zs) with zs = -z*s, rounded to T, natural k order.
template <typename T>
__device__ __forceinline__ uint4 dequant_word(uint32_t w, float s, float zs) {
  const uint32_t t0 = w & 0x0F0F0F0Fu;         // bytes: k0, k4, k1, k5
  const uint32_t t1 = (w >> 4) & 0x0F0F0F0Fu;  // bytes: k2, k6, k3, k7
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t k01 = {cvt_ubyte0(t0), cvt_ubyte2(t0)};
  f32x2_t k23 = {cvt_ubyte0(t1), cvt_ubyte2(t1)};
  f32x2_t k45 = {cvt_ubyte1(t0), cvt_ubyte3(t0)};
  f32x2_t k67 = {cvt_ubyte1(t1), cvt_ubyte3(t1)};
  // eight v_fma_f32, not four v_pk_fma_f32: next to MFMAs the packed form is the slower one
  // (scripts/ubench/mfma_valu_overlap.hip: 128 v_pk_fma_f32 + 32 MFMAs per SIMD take 1412 cycles,
  // more than the 549 + 609 they take apart; 128 v_fma_f32 + 32 MFMAs 819) — gate_up at M = 64
  // 36.5 -> 34.5 us.  Inline asm: the vectoriser would fuse the pairs back into v_pk_fma_f32.
  auto f = [&](float q) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(s), "v"(zs));
    return r;
  };
  k01 = f32x2_t{f(k01.x), f(k01.y)};
  k23 = f32x2_t{f(k23.x), f(k23.y)};
  k45 = f32x2_t{f(k45.x), f(k45.y)};
  k67 = f32x2_t{f(k67.x), f(k67.y)};
  uint4 r;
  r.x = Mfma<T>::pack(k01.x, k01.y);
  r.y = Mfma<T>::pack(k23.x, k23.y);
  r.z = Mfma<T>::pack(k45.x, k45.y);
  r.w = Mfma<T>::pack(k67.x, k67.y);
  return r;
}

// The same value from half as many converts: a byte 0x0q read as OCP e4m3 is q * 2^-9 for every q in 0..15 (the
// nibble's top bit lands in the exponent field, the denormal range continues linearly into the first binade), so one
// v_cvt_pk_f32_fp8 turns two masked nibbles into two floats.  s512 = 512 * s: (q 2^-9) (512 s) is the same real
// number as q s, the fma rounds once — bit-identical to dequant_word (scripts/ubench/valu_rate.hip prints the
// identity and the issue rates).
template <typename T>
__device__ __forceinline__ uint4 dequant_word_fp8pk(uint32_t w, float s512, float zs) {
  const uint32_t t0 = w & 0x0F0F0F0Fu;         // bytes: k0, k4, k1, k5
  const uint32_t t1 = (w >> 4) & 0x0F0F0F0Fu;  // bytes: k2, k6, k3, k7
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t k04, k15, k26, k37;
  asm("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_0" : "=v"(k04) : "v"(t0));
  asm("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(k15) : "v"(t0));
  asm("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_0" : "=v"(k26) : "v"(t1));
  asm("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(k37) : "v"(t1));
  auto f = [&](float q) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(s512), "v"(zs));
    return r;
  };
  uint4 r;
  r.x = Mfma<T>::pack(f(k04.x), f(k15.x));
  r.y = Mfma<T>::pack(f(k26.x), f(k37.x));
  r.z = Mfma<T>::pack(f(k04.y), f(k15.y));
  r.w = Mfma<T>::pack(f(k26.y), f(k37.y));
  return r;
}

enum ZeroMode { kZeroAwq = 0, kZeroGptq = 1 };

// zero points of the 4 columns n .. n+3 (n % 4 == 0) from their packed word
template <int ZMODE>
__device__ __forceinline__ void unpack_zeros4(uint32_t w, int n, float (&z)[4]) {
  const int base = n & 7;  // 0 or 4
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (ZMODE == kZeroAwq) {
      // natural column j sits at nibble awq_shift(j): {0,4,1,5} for base 0, {2,6,3,7} for base 4
      const int sh = 4 * ((base >> 1) + (t >> 1) + 4 * (t & 1));
      z[t] = (float)((w >> sh) & 0xFu);
    } else {
      z[t] = (float)(((w >> (4 * (base + t))) & 0xFu) + 1u);
    }
  }
}
template <int ZMODE>
__device__ __forceinline__ void load_zeros4(const uint32_t* __restrict__ zrow, int n, float (&z)[4]) {
  unpack_zeros4<ZMODE>(zrow[n >> 3], n, z);
}

// slot of lane-column lc, k-group lr inside a 1 KiB MFMA operand image (see w4a16_unfused.hip)
__device__ __forceinline__ int frag_swz(int lr, int lc) {
  // g = {0, 12, 2, 14}[lr]  ==  ((lr & 1) * 12) | ((lr & 2))
  return lr * 16 + (lc ^ (((lr & 1) * 12) | (lr & 2)));
}

// ---- pack activations into operand images ------------------------------------------------
// A [M, K] row-major -> pieces PA[mt][kt][64 slots x 16 B] in the same swizzled operand image
// (slot swz(lr, lc) = A[16 mt + lc][32 kt + 8 lr .. +7]); rows >= M are zero.  One workgroup
// packs 16 rows x kPackK k: coalesced 16-B reads (a row's 2 KiB by 128 lanes), all 8 loads of a
// thread in flight before the shuffle in LDS, the 32 KiB image is written out linearly.
// interleave != 0: piece 4 G + t holds rows 64 G + 4 r + t (r = 0..15) instead of 16 consecutive
// rows, so that a lane of the consuming GEMM ends up with 4 adjacent outputs along this axis.
constexpr int kPackK = 1024;
template <typename T>
__global__ __launch_bounds__(256) void pack_a_kernel(T* __restrict__ packed, const T* __restrict__ a,
                                                     int m, int k, int64_t lda,
                                                     int interleave = 0) {
  __shared__ uint4 img[(kPackK / 32) * 64];        // 32 pieces of 1 KiB
  const int mt = blockIdx.y;
  const int k0 = blockIdx.x * kPackK;              // first k of this block
  const int kt32 = k >> 5;
  constexpr int kChunksPerRow = kPackK / 8;        // 16-B chunks per row
  constexpr int kIters = 16 * kChunksPerRow / 256; // 8
  uint4 v[kIters];
#pragma unroll
  for (int i = 0; i < kIters; ++i) {
    const int idx = threadIdx.x + i * 256;         // row = idx / 128, chunk = idx % 128
    const int r = idx / kChunksPerRow, ch = idx % kChunksPerRow;
    const int row = interleave ? (mt >> 2) * 64 + 4 * r + (mt & 3) : mt * 16 + r;
    const int kk = k0 + ch * 8;
    v[i] = make_uint4(0, 0, 0, 0);
    if (row < m && kk < k) v[i] = *reinterpret_cast<const uint4*>(a + (int64_t)row * lda + kk);
  }
#pragma unroll
  for (int i = 0; i < kIters; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int r = idx / kChunksPerRow, ch = idx % kChunksPerRow;
    img[(ch >> 2) * 64 + frag_swz(ch & 3, r)] = v[i];
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(packed) + ((int64_t)mt * kt32 + (k0 >> 5)) * 64;
  const int pieces = min(kPackK / 32, kt32 - (k0 >> 5));
  for (int idx = threadIdx.x; idx < pieces * 64; idx += 256) dst[idx] = img[idx];
}

// K split of the 256 x 256-tile prefill kernels (gemm_packed_kernel, gemm8_packed_kernel) for shapes whose tiles leave
// most of the chip idle (chunked-prefill sized M, the narrow per-rank N of a TP = 8 shard): sk K ranges per tile,
// partial tiles through 4-byte slabs [sk][m][n] and a finish kernel that adds them in split order.  steps = the
// tile loop's length in 128-byte k-steps (fp8 / int8: k / 128; 2-byte operands: k / 64).  Cost model in us (fitted
// to scripts/bench_scaled_mm.py: ~1.6 us per k-step, ~8 us of prologue + epilogue, slabs written and read back at
// ~5 TB/s):  t(sk) = rounds(tiles * sk) * (ceil(steps / sk) * 1.6 + 8) + [sk > 1] * (4 + 8 sk m n / 5e6).
// Every split keeps >= 4 k-steps and is non-empty; `max_slab_elems` bounds sk by the workspace; forced > 0: that sk
// (or 1 where it is not admissible).
constexpr int kW4PrepackedMinM = 384;   // smallest m the operand-image GEMM takes when the weights' image exists
static inline int plan_tile_split(int m, int n, int steps, int64_t max_slab_elems, int forced) {
  const int tiles = ((m + 255) / 256) * ((n + 255) / 256);
  if (tiles >= 160 || steps < 8) return 1;
  int best = 1;
  double best_t = 1e30;
  for (int sk = 1; sk <= 8; ++sk) {
    const int per = (steps + sk - 1) / sk;
    if (sk > 1 && (per < 4 || (int64_t)per * (sk - 1) >= steps || (int64_t)sk * m * n > max_slab_elems)) continue;
    if (forced > 0 && sk != forced && sk != 1) continue;
    const int rounds = (tiles * sk + 255) / 256;
    double t = rounds * (per * 1.6 + 8.0);
    if (sk > 1) t += 4.0 + 8.0 * sk * (double)m * n / 5e6;
    if (forced > 0 && sk == forced) { best = sk; break; }
    if (t < best_t) { best_t = t; best = sk; }
  }
  return best;
}

// out = T(slab[0] + slab[1] + ... + slab[sk-1])  (fixed order: results do not depend on timing)
template <typename T>
__global__ void w4a16_sum_slabs_kernel(T* __restrict__ out, const float* __restrict__ slabs,
                                       int64_t n4, int sk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float v[4];
  sum_slabs<4>(slabs + i * 4, sk, n4 * 4, v);
  T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
  reinterpret_cast<uint2*>(out)[i] = *reinterpret_cast<const uint2*>(o);
}

struct GemmArgs {
  void* c;
  const void* a;
  const uint32_t* qw;
  const void* scales;
  const uint32_t* qz;
  float* ws;
  int64_t ws_elems;
  void* dq_ws;           // optional scratch for the dequantised weights (unfused prefill path)
  int64_t dq_ws_bytes;
  int m, n, k, group;
  int64_t lda;
  int zmode;
  hipStream_t stream;
  int* defer_sk = nullptr;   // non-null: leave split-K slabs unreduced, report their count here
  bool fuse_silu = false;    // prefill gate_up: write silu_and_mul(C) [m, n/2] instead of C
  bool out_packed = false;   // with fuse_silu, m >= 1024: write it as the operand image of the next GEMM
  bool a_packed = false;     // m >= 1024: `a` already is an operand image (pack_a_kernel's format)
  int bits = 4;              // 8: GPTQ 8-bit (words [K/4][N], byte i of word (kk, n) = W[4kk + i][n], qzeros
                             // [K/g][N/4], zero = qzeros + 1; q_gemm.cu:1998-2003 kU8B128 on symmetric files)
  const void* b_image = nullptr;   // m >= 1024: the weights' operand image, dequantised once at load time
                                   // (mi355x_w4a16_prepack); qw / scales / qz are then unused
};


}  // namespace mi355x
