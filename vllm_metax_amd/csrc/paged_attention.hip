// paged_attention.hip — single-query (decode) attention over the paged KV cache,
// hand-written for gfx950 (wave64, v_dot2c_f32_{bf16,f16}, DPP/permlane shuffles).
//
// Reference semantics restated (order of operations and rounding points):
//   csrc/attention/attention_kernels.cuh:75-485  (paged_attention_kernel)
//     logits fp32 = scale * <q,k> (+ alibi_slope * (tok - seq_len + 1)); masked
//     tokens excluded from the max; p = exp(l - max) * 1/(sum + 1e-6);
//     p rounded to scalar_t BEFORE the PV product; PV accumulated in fp32;
//     tail-block V lanes zeroed.
//   :519-551 (v2 partition wrapper: PARTITION_SIZE = 512, per-partition max / sum,
//     tmp_out in scalar_t), :553-658 (reduce: exp_sums[j]*exp(max_j-max), 1/(S+1e-6)).
//   csrc/attention/paged_attention_v1.cu:43-125, paged_attention_v2.cu:43-131.
//
// MI355X design (not the reference's):
//   * grid = (kv_head x head_tile, seq, partition); ONE workgroup serves up to GT=4
//     query heads of a KV head, so every K/V byte is read from HBM once per GQA
//     group instead of once per query head (4x fewer bytes on Llama-3).
//   * the x-split layout [blk, kvh, d/x, bs, x] makes every 16-token K block a
//     contiguous 4 KiB run: a wave fetches it with 16-B-per-lane loads whose 64
//     lanes cover 1 KiB contiguous (lane = chunk_sub*BS + token), straight into
//     VGPRs; V [blk, kvh, d, bs] likewise.  Two blocks are kept in flight per wave.
//   * QK^T and PV are v_dot2c_f32_bf16 / _f16 on packed pairs (fp32 accumulate);
//     the probabilities are stored in LDS already rounded to scalar_t, which is
//     exactly the reference's rounding point.
//   * reductions across the lanes that share a token are wave shuffles; across
//     waves through LDS.
//   * fp8 (e4m3fn) KV cache (kv_cache_dtype "fp8"; hook points of the reference:
//     attention_kernels.cuh:86-89 cache_t, :266-277 / :398-407 scaled_convert; its dispatch rejects
//     it, quant_utils.cuh:29-42 — semantics are upstream vLLM's): the cache holds bytes, x = 16, so
//     a 16-B piece is 16 head elements of one token (K) / 16 tokens of one row (V).  A piece is
//     converted ONCE to scalar_t in registers (exact: e4m3 is a subset of bf16 / f16 / f32) and
//     then used for all GT query heads; k_scale is folded into the logit scale and v_scale into the
//     output (upstream rounds T(byte * scale) per element first: identical for scale 1 and for
//     powers of two, otherwise this form carries one rounding less).
#include <type_traits>

#include "common.cuh"

namespace mi355x {

constexpr int kPaThreads = 256;       // default workgroup: 4 waves
constexpr int kPaMaxThreads = 512;    // few-workgroup launches (v1 at small batch) use 8 waves
constexpr int pa_scratch_bytes(int gt) { return gt > 4 ? 512 : 256; }  // red[GT][16] floats

// 16-byte load of a K / V piece: non-temporal — the cache is streamed once per step, keeping it out of
// L2 / MALL is worth 10 % (64 seqs x ctx 1088, v1: 54.3 -> 48.5 us, fp8 cache 34.8 -> 30.5 us;
// profiles/r02_attn_ab2.txt).  -DPA_NO_NT builds the default-policy variant of that comparison.
__device__ __forceinline__ uint4 ld_kv16(const void* p) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Block-wide max / sum of N values at once: ONE pair of barriers for all of them (the softmax of the
// GT query heads of a workgroup used to pay two barriers per head and reduction).
template <bool IS_MAX, int N>
__device__ __forceinline__ void block_reduce_n(float (&v)[N], float* smem /* [N][16] */) {
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int g = 0; g < N; ++g) v[g] = IS_MAX ? wave_max(v[g]) : wave_sum(v[g]);
  if (nw == 1) return;
  __syncthreads();  // protect smem from a previous use
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < N; ++g) smem[g * 16 + wid] = v[g];
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < N; ++g) {
    float r = IS_MAX ? -3.402823466e+38f : 0.f;
    if (lane < nw) r = smem[g * 16 + lane];
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      const float o = __shfl_xor(r, m, 64);
      r = IS_MAX ? fmaxf(r, o) : r + o;
    }
    v[g] = __shfl(r, 0, 64);
  }
}

template <typename T>
__device__ __forceinline__ float dot_chunk(const uint4& a, const uint4& b, float acc);

template <>
__device__ __forceinline__ float dot_chunk<bf16_t>(const uint4& a, const uint4& b, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x),
                                        __builtin_bit_cast(bf16x2_t, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y),
                                        __builtin_bit_cast(bf16x2_t, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z),
                                        __builtin_bit_cast(bf16x2_t, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w),
                                        __builtin_bit_cast(bf16x2_t, b.w), acc, false);
  return acc;
}
template <>
__device__ __forceinline__ float dot_chunk<f16_t>(const uint4& a, const uint4& b, float acc) {
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.x), __builtin_bit_cast(f16x2_t, b.x),
                               acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.y), __builtin_bit_cast(f16x2_t, b.y),
                               acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.z), __builtin_bit_cast(f16x2_t, b.z),
                               acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.w), __builtin_bit_cast(f16x2_t, b.w),
                               acc, false);
  return acc;
}
template <>
__device__ __forceinline__ float dot_chunk<float>(const uint4& a, const uint4& b, float acc) {
  acc = fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
  acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
  acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
  acc = fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
  return acc;
}

// 16 e4m3 (E5M2: e5m2) bytes -> 16 scalar_t values (QP = 16 * sizeof(T) / 16 uint4s), exact.
template <typename T, bool E5M2 = false>
struct Fp8Piece;
template <bool E5M2>
struct Fp8Piece<bf16_t, E5M2> {
  static constexpr int QP = 2;
  static __device__ __forceinline__ void cvt(const uint4& r, uint4 (&t)[2]) {
    auto lo = [](uint32_t w) {
      if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, 1.0f, false));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    };
    auto hi = [](uint32_t w) {
      if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, 1.0f, true));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true));
    };
    t[0] = make_uint4(lo(r.x), hi(r.x), lo(r.y), hi(r.y));
    t[1] = make_uint4(lo(r.z), hi(r.z), lo(r.w), hi(r.w));
  }
};
template <bool E5M2>
struct Fp8Piece<f16_t, E5M2> {
  static constexpr int QP = 2;
  static __device__ __forceinline__ void cvt(const uint4& r, uint4 (&t)[2]) {
    auto lo = [](uint32_t w) {
      if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w, 1.0f, false));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false));
    };
    auto hi = [](uint32_t w) {
      if constexpr (E5M2) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w, 1.0f, true));
      else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true));
    };
    t[0] = make_uint4(lo(r.x), hi(r.x), lo(r.y), hi(r.y));
    t[1] = make_uint4(lo(r.z), hi(r.z), lo(r.w), hi(r.w));
  }
};
template <bool E5M2>
struct Fp8Piece<float, E5M2> {
  static constexpr int QP = 4;
  static __device__ __forceinline__ void cvt(const uint4& r, uint4 (&t)[4]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 a, b;
      if constexpr (E5M2) {
        a = __builtin_amdgcn_cvt_pk_f32_bf8(w[i], false);
        b = __builtin_amdgcn_cvt_pk_f32_bf8(w[i], true);
      } else {
        a = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false);
        b = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
      }
      t[i] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x),
                        __float_as_uint(b.y));
    }
  }
};

// One 16-B cache piece as scalar_t values: CT == T -> the piece itself; CT == uint8_t -> converted.
template <typename T, typename CT>
struct Piece {
  static constexpr bool kE5M2 = std::is_same<CT, e5m2_t>::value;
  static constexpr int QP = std::is_same<T, CT>::value ? 1 : Fp8Piece<T, kE5M2>::QP;
  static __device__ __forceinline__ void cvt(const uint4& r, uint4 (&t)[QP]) {
    if constexpr (std::is_same<T, CT>::value) t[0] = r;
    else Fp8Piece<T, kE5M2>::cvt(r, t);
  }
  static __device__ __forceinline__ float dot(const uint4 (&a)[QP], const uint4 (&b)[QP], float acc) {
#pragma unroll
    for (int p = 0; p < QP; ++p) acc = dot_chunk<T>(a[p], b[p], acc);
    return acc;
  }
};

// zero the elements of a 16-B V piece whose token index is >= seq_len (a zero byte is +0 in e4m3)
template <typename CT>
__device__ __forceinline__ uint4 mask_tail(uint4 v, int first_token, int seq_len) {
  constexpr int X = 16 / sizeof(CT);
  CT e[X];
  *reinterpret_cast<uint4*>(e) = v;
#pragma unroll
  for (int j = 0; j < X; ++j) {
    if (first_token + j >= seq_len) e[j] = CT(0);
  }
  return *reinterpret_cast<uint4*>(e);
}

// FQ (mi355x_paged_attention_fused_qkv): the decode step's "slab sum -> NeoX rotary -> reshape_and_cache" of
// the new token happens in THIS kernel's prologue instead of a launch of its own (qkv_rope_cache_kernel, the
// same arithmetic): the workgroup of (sequence, kv head) builds its GT query heads, its k head and its v head
// from the qkv row (or its split-K slabs), rotates q and k, keeps q in LDS and writes k / v into the cache slot
// of the new token, which its own QK^T / PV loops then read back (same workgroup: a release fence + barrier
// orders the stores before the loads; no other workgroup reads that kv head of that sequence).
struct FusedQkv {
  const void* qkv;            // [num_seqs, qkv_stride] scalar_t rows q | k | v (read when sk == 0)
  int64_t qkv_stride;
  const float* slabs;         // [sk][num_seqs][width] fp32 split-K partials of the qkv projection, or null
  int sk;
  int64_t slab_stride;
  const int64_t* positions;   // [num_seqs]
  const void* cos_sin_cache;  // [max_pos, head_size] scalar_t
  const int64_t* slot_mapping;  // [num_seqs]
  SlabScales scales;            // fp8 qkv GEMM: the slabs still want a_scale[token] * b_scale[column] (common.cuh)
};

// HS == 0: head size is a run-time value (any multiple of 16/sizeof(T) up to 256).
// CT: element type of the cache (T, or uint8_t = e4m3fn bytes with x = 16).
template <typename T, typename CT, int BS, int GT, int HS, bool FQ = false>
__global__ __launch_bounds__(kPaMaxThreads) void paged_attention_kernel(
    float* __restrict__ exp_sums,    // [num_seqs, num_heads, P]       (partitioned only)
    float* __restrict__ max_logits,  // [num_seqs, num_heads, P]       (partitioned only)
    T* __restrict__ out,             // [num_seqs, num_heads, P, head_size]
    const T* __restrict__ q,         // [num_seqs, num_heads, head_size]
    // (FQ: the kernel writes the caches too — no __restrict__ promise on them)
    typename std::conditional<FQ, const CT*, const CT* __restrict__>::type k_cache,  // [num_blocks, num_kv_heads, head_size/x, BS, x]
    typename std::conditional<FQ, const CT*, const CT* __restrict__>::type v_cache,  // [num_blocks, num_kv_heads, head_size, BS]
    int num_heads, int num_kv_heads, int head_size_rt, float scale,
    const int* __restrict__ block_tables, const int* __restrict__ seq_lens,
    int max_num_blocks_per_seq, const float* __restrict__ alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int partition_size, int logits_cap,
    const float* __restrict__ k_scale, const float* __restrict__ v_scale, FusedQkv fq = FusedQkv{}) {
  constexpr bool KV8 = !std::is_same<T, CT>::value;
  static_assert(!FQ || (!KV8 && HS != 0 && sizeof(T) == 2), "fused qkv: 2-byte scalar_t cache, fixed head size");
  constexpr int XT = 16 / sizeof(T);           // scalar_t elements per 16 B (q staging)
  constexpr int X = 16 / sizeof(CT);           // cache elements per 16-B piece
  constexpr int QP = Piece<T, CT>::QP;         // uint4s of scalar_t that one piece expands to
  constexpr int LPT = 64 / BS;                 // lanes that share one token in QK
  constexpr int TPP = BS / X;                  // 16-B pieces per V row
  static_assert(TPP >= 1, "block_size must cover at least one 16-byte piece of the cache type");
  constexpr int DPI = 64 / TPP;                // V rows covered by one wave load
  const int D = HS ? HS : head_size_rt;
  const int C = D / X;                         // 16-B pieces per head vector in the cache
  const int CQ = D / XT;                       // 16-B chunks of a query row
  constexpr int NI = HS ? (HS / X + LPT - 1) / LPT : 0;   // K pieces per lane per block
  constexpr int NIV = HS ? (HS + DPI - 1) / DPI : (256 + DPI - 1) / DPI;
  if constexpr (KV8) scale *= *k_scale;        // logits = scale * k_scale * <q, K8>

  const int seq = blockIdx.y;
  const int part = blockIdx.z;
  const int num_parts = gridDim.z;
  const int seq_len = seq_lens[seq];
  const bool partitioned = partition_size > 0;
  if (partitioned && part * partition_size >= seq_len) return;

  const int q_per_kv = num_heads / num_kv_heads;
  const int tiles = (q_per_kv + GT - 1) / GT;
  const int kv_head = blockIdx.x / tiles;
  const int tile = blockIdx.x - kv_head * tiles;
  const int head0 = kv_head * q_per_kv + tile * GT;       // first query head of this WG
  const int nheads = min(GT, q_per_kv - tile * GT);       // valid heads in the tile

  const int num_seq_blocks = (seq_len + BS - 1) / BS;
  const int blocks_per_part = partitioned ? partition_size / BS : num_seq_blocks;
  const int start_block = partitioned ? part * blocks_per_part : 0;
  const int end_block = min(start_block + blocks_per_part, num_seq_blocks);
  const int start_token = start_block * BS;
  const int num_tokens = min(start_token + (end_block - start_block) * BS, seq_len) - start_token;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: block ids / bases on the SALU
  const int nthreads = blockDim.x;           // 256 or 512 (host: launch_pa)
  const int nwaves = nthreads >> 6;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // all LDS lives in the dynamic region so that its base stays 16-B aligned
  float* red = reinterpret_cast<float*>(smem);                          // [GT][16]
  constexpr int kPaScratchBytes = pa_scratch_bytes(GT);
  T* q_s = reinterpret_cast<T*>(smem + kPaScratchBytes);                // [GT][256]
  float* logits =
      reinterpret_cast<float*>(smem + kPaScratchBytes + (size_t)GT * 256 * sizeof(T));  // [GT][cap]
  T* probs = reinterpret_cast<T*>(logits + (size_t)GT * logits_cap);    // [GT][cap]
  // physical block ids of this (sequence, partition): read once, coalesced, instead of one dependent
  // global load in front of every K / V block
  int* bt_s = reinterpret_cast<int*>(probs + (size_t)GT * logits_cap);  // [cap / BS]

  // physical block ids of the partition first: their latency overlaps with the q staging / prologue below
  const int* block_table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  for (int i = tid; i < end_block - start_block; i += nthreads) bt_s[i] = block_table[start_block + i];
  // Work split: block (start_block + i * nwaves + wave) is the i-th item of this wave (round robin:
  // the waves' loads differ by at most one block).  K and V blocks go through TWO register buffers:
  // the load of item i + 2 is issued as soon as item i has been consumed, so one to two 4-KiB blocks
  // per wave (64-128 KiB per CU) are in flight all the time instead of "load two, wait, compute two".
  const int t_in_blk = lane % BS;     // token of this lane inside a block
  const int csub = lane / BS;         // chunk phase of this lane
  const int nblk_part = end_block - start_block;
  const int nitems = wave < nblk_part ? (nblk_part - wave + nwaves - 1) / nwaves : 0;
  auto item_block = [&](int i) { return i * nwaves + wave; };     // index inside the partition
  constexpr int NIK = NI > 0 ? NI : 1;
  uint4 ka[NIK], kb[NIK];
  auto load_k = [&](int i, uint4 (&kk)[NIK]) {
    const int64_t pb = bt_s[item_block(i)];
    const CT* kp = k_cache + pb * kv_block_stride + (int64_t)kv_head * kv_head_stride;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c = csub + LPT * j;
      const int cc = (c < C) ? c : 0;
      kk[j] = ld_kv16(kp + (cc * BS + t_in_blk) * X);
    }
  };
  bool pre0 = false, pre1 = false;    // FQ: K blocks 0 / 1 of this wave were requested inside the prologue
  if constexpr (FQ) {
    // ---- q | k | v of the new token: [slab sum ->] T, rotary on q and k, k / v into the cache ----
    // (host: tiles == 1 and nheads == GT; rows of q_s: GT query heads, then the k head, then the v head)
    const int width = (num_heads + 2 * num_kv_heads) * D;
    // position, slot and this thread's cos / sin chunk are requested up front, with the slab loads: behind
    // them they would be two more dependent memory round trips in front of the first K load
    const int embed = D / 2;
    const int cph = embed / XT;
    const int64_t slot = fq.slot_mapping[seq];
    // (cos_sin_cache == nullptr: q and k arrive already rotated — the backend form, where the model's rotary ran
    //  before the attention layer: only the cache write of the new token is folded in)
    const bool rotate = fq.cos_sin_cache != nullptr;
    const T* cos_ptr = rotate ? static_cast<const T*>(fq.cos_sin_cache) + fq.positions[seq] * D : nullptr;
    Vec16<T> cs0, sn0;
    if (rotate && tid < (GT + 1) * cph) {
      cs0 = load16(cos_ptr + (tid % cph) * XT);
      sn0 = load16(cos_ptr + embed + (tid % cph) * XT);
    }
    for (int i = tid; i < (GT + 2) * CQ; i += nthreads) {
      const int hs = i / CQ;
      const int c = i - hs * CQ;
      const int col = (hs < GT ? head0 + hs : (hs == GT ? num_heads + kv_head : num_heads + num_kv_heads + kv_head)) * D + c * XT;
      Vec16<T> v;
      if (fq.sk > 0) {
        slab_values<T, XT>(fq.slabs + (int64_t)seq * width + col, fq.sk, fq.slab_stride, fq.scales, seq, col, v.e);
      } else {
        v = load16(static_cast<const T*>(fq.qkv) + (int64_t)seq * fq.qkv_stride + col);
      }
      store16(q_s + hs * D + c * XT, v);
    }
    __syncthreads();   // (also publishes bt_s)
    {
      // the first two K blocks of this wave go out NOW — unless one of them is the block that receives the new
      // token (written below): their latency overlaps with the rotary, the cache write and its fence
      const int new_local = (seq_len - 1) / BS - start_block;
      pre0 = nitems > 0 && item_block(0) != new_local;
      pre1 = nitems > 1 && item_block(1) != new_local;
      if (pre0) load_k(0, ka);
      if (pre1) load_k(1, kb);
    }
    if (rotate && tid < (GT + 1) * cph) {     // (GT + 1) * cph <= 72 <= the workgroup size: one chunk pair per thread
      const int h = tid / cph;
      const int c = tid - h * cph;
      T* base = q_s + h * D;
      Vec16<T> x = load16(base + c * XT);
      Vec16<T> y = load16(base + embed + c * XT);
#pragma unroll
      for (int j = 0; j < XT; ++j) rot_pair<T>(x.e[j], y.e[j], cs0.e[j], sn0.e[j]);
      store16(base + c * XT, x);
      store16(base + embed + c * XT, y);
    }
    __syncthreads();
    // the workgroup whose token range ends with the new token (v1: the only one) writes it
    if (slot >= 0 && (!partitioned || part == (seq_len - 1) / partition_size)) {
      const int64_t blk = slot / BS;
      const int t = (int)(slot - blk * BS);
      T* kw = const_cast<T*>(reinterpret_cast<const T*>(k_cache)) + blk * kv_block_stride + (int64_t)kv_head * kv_head_stride;
      T* vw = const_cast<T*>(reinterpret_cast<const T*>(v_cache)) + blk * kv_block_stride + (int64_t)kv_head * kv_head_stride;
      for (int c = tid; c < CQ; c += nthreads) store16(kw + (c * BS + t) * XT, load16(q_s + GT * D + c * XT));
      for (int d = tid; d < D; d += nthreads) vw[(int64_t)d * BS + t] = q_s[(GT + 1) * D + d];
    }
    // the stores are complete before the barrier below (its vmcnt(0) also drains the prefetched K blocks: a
    // timing-only build without the fence ran the in-job launch in 54.0-54.5 us against 54.4-54.6 — nothing to gain,
    // profiles/r03_attn_fq_nofence.txt)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  } else {
  // ---- stage q (packed scalar_t) into LDS; absent heads are zero ---------------
  for (int i = tid; i < GT * CQ; i += nthreads) {
    const int g = i / CQ;
    const int c = i - g * CQ;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g < nheads) {
      v = *reinterpret_cast<const uint4*>(q + (int64_t)seq * q_stride + (int64_t)(head0 + g) * D + c * XT);
    }
    *reinterpret_cast<uint4*>(q_s + g * D + c * XT) = v;
  }
  }
  __syncthreads();
  if constexpr (FQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  float slope[GT];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    slope[g] = (alibi_slopes != nullptr && g < nheads) ? alibi_slopes[head0 + g] : 0.f;
  }

  // =========================== QK^T ==============================================
  float qk_max[GT];
#pragma unroll
  for (int g = 0; g < GT; ++g) qk_max[g] = -3.402823466e+38f;

  if constexpr (HS != 0) {
    // q chunks of this lane live in registers for the whole kernel — up to 4 heads per workgroup.  GT = 8 (one
    // workgroup for ALL 8 query heads of a kv head: the TP = 8 shards of Llama-3-70B / Qwen2-72B, 8 q / 1 kv head)
    // would need 128 VGPRs for them: there the chunks are re-read from LDS per K piece (the lanes of a token
    // group read one address: a broadcast, 8 ds_read_b128 per 16-byte piece of K against 32 v_dot2c).
    constexpr bool QLDS = GT > 4;
    uint4 qreg[QLDS ? 1 : GT][NI][QP];
    if constexpr (!QLDS) {
#pragma unroll
      for (int g = 0; g < GT; ++g) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int c = csub + LPT * i;
#pragma unroll
          for (int p = 0; p < QP; ++p) {
            qreg[g][i][p] = (c < C) ? *reinterpret_cast<const uint4*>(q_s + g * D + c * X + p * XT)
                                    : make_uint4(0, 0, 0, 0);
          }
        }
      }
    }
    auto qk_block = [&](int i, const uint4 (&kk)[NIK]) {
      float acc[GT];
#pragma unroll
      for (int g = 0; g < GT; ++g) acc[g] = 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        uint4 t[QP];                      // the piece as scalar_t (converted once, used by GT heads)
        Piece<T, CT>::cvt(kk[j], t);
        if constexpr (QLDS) {
          const int c = csub + LPT * j;
          if (c < C) {
#pragma unroll
            for (int g = 0; g < GT; ++g) {
              uint4 qv[QP];
#pragma unroll
              for (int p = 0; p < QP; ++p) qv[p] = *reinterpret_cast<const uint4*>(q_s + g * D + c * X + p * XT);
              acc[g] = Piece<T, CT>::dot(t, qv, acc[g]);
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < GT; ++g) acc[g] = Piece<T, CT>::dot(t, qreg[g][j], acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < GT; ++g) {
#pragma unroll
        for (int m = BS; m < 64; m <<= 1) acc[g] += __shfl_xor(acc[g], m, 64);
      }
      if (csub == 0) {
        const int loc = item_block(i) * BS + t_in_blk;        // token index inside the partition
        const int tok = start_token + loc;
        const bool msk = tok >= seq_len;
#pragma unroll
        for (int g = 0; g < GT; ++g) {
          float v = acc[g] * scale;
          v += (slope[g] != 0.f) ? slope[g] * (tok - seq_len + 1) : 0.f;
          logits[g * logits_cap + loc] = msk ? 0.f : v;
          qk_max[g] = msk ? qk_max[g] : fmaxf(qk_max[g], v);
        }
      }
    };
    if (nitems > 0 && !pre0) load_k(0, ka);
    if (nitems > 1 && !pre1) load_k(1, kb);
    for (int i = 0; i < nitems; i += 2) {
      qk_block(i, ka);
      if (i + 2 < nitems) load_k(i + 2, ka);
      if (i + 1 < nitems) {
        qk_block(i + 1, kb);
        if (i + 3 < nitems) load_k(i + 3, kb);
      }
    }
  } else {
    for (int blk = start_block + wave; blk < end_block; blk += nwaves) {
      const int64_t pb = bt_s[blk - start_block];
      const CT* kp = k_cache + pb * kv_block_stride + (int64_t)kv_head * kv_head_stride;
      float a[GT];
#pragma unroll
      for (int g = 0; g < GT; ++g) a[g] = 0.f;
      for (int c = csub; c < C; c += LPT) {
        const uint4 kv = *reinterpret_cast<const uint4*>(kp + (c * BS + t_in_blk) * X);
        uint4 kt[QP];
        Piece<T, CT>::cvt(kv, kt);
#pragma unroll
        for (int g = 0; g < GT; ++g) {
          uint4 qv[QP];
#pragma unroll
          for (int p = 0; p < QP; ++p) qv[p] = *reinterpret_cast<const uint4*>(q_s + g * D + c * X + p * XT);
          a[g] = Piece<T, CT>::dot(kt, qv, a[g]);
        }
      }
      const int tok = blk * BS + t_in_blk;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
#pragma unroll
        for (int m = BS; m < 64; m <<= 1) a[g] += __shfl_xor(a[g], m, 64);
        if (csub == 0) {
          float v = a[g] * scale;
          v += (slope[g] != 0.f) ? slope[g] * (tok - seq_len + 1) : 0.f;
          const bool msk = tok >= seq_len;
          logits[g * logits_cap + tok - start_token] = msk ? 0.f : v;
          qk_max[g] = msk ? qk_max[g] : fmaxf(qk_max[g], v);
        }
      }
    }
  }

  // The first two V blocks of this wave are requested BEFORE the softmax: nothing in the softmax
  // touches global memory, so these loads (and the HBM latency behind them) overlap with it.
  constexpr int TPP0 = BS / X;
  constexpr int DPI0 = 64 / TPP0;
  const int tq = lane % TPP0;     // which 16-B piece (X tokens) of a V row
  const int dsub = lane / TPP0;   // V row phase
  auto load_v = [&](int blk, uint4 (&vv)[NIV]) {
    const int64_t pb = bt_s[blk - start_block];
    const CT* vp = v_cache + pb * kv_block_stride + (int64_t)kv_head * kv_head_stride;
    // (the tail of the last block is masked in pv_block, in ONE branch behind all the loads: a test here made
    //  hipcc branch around every load with a full vmcnt(0) behind it — the NIV loads of a block, and the two
    //  prefetched blocks in front of the softmax, then went to memory one after the other)
#pragma unroll
    for (int i = 0; i < NIV; ++i) {
      const int d = dsub + DPI0 * i;
      vv[i] = (d < D) ? ld_kv16(vp + (int64_t)d * BS + tq * X) : make_uint4(0, 0, 0, 0);
    }
  };
  uint4 vpre0[NIV], vpre1[NIV];
  if constexpr (HS != 0) {
    if (nitems > 0) load_v(start_block + item_block(0), vpre0);
    if (nitems > 1) load_v(start_block + item_block(1), vpre1);
  }

  // =========================== softmax ===========================================
  block_reduce_n<true, GT>(qk_max, red);          // every thread now holds the GT maxima
  float lsum[GT];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    lsum[g] = 0.f;
    const float m = qk_max[g];
    for (int i = tid; i < num_tokens; i += nthreads) {
      const float e = __expf(logits[g * logits_cap + i] - m);
      logits[g * logits_cap + i] = e;
      lsum[g] += e;
    }
  }
  block_reduce_n<false, GT>(lsum, red);           // (its first barrier also orders the logits writes)
  const int padded_tokens = (end_block - start_block) * BS;
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    const float inv = __fdividef(1.f, lsum[g] + 1e-6f);
    for (int i = tid; i < padded_tokens; i += nthreads) {
      const float p = (i < num_tokens) ? logits[g * logits_cap + i] * inv : 0.f;
      probs[g * logits_cap + i] = from_f32<T>(p);
    }
  }
  if (partitioned && tid == 0) {
#pragma unroll
    for (int g = 0; g < GT; ++g) {
      if (g < nheads) {
        const int64_t o = ((int64_t)seq * num_heads + head0 + g) * num_parts + part;
        max_logits[o] = qk_max[g];
        exp_sums[o] = lsum[g];
      }
    }
  }
  __syncthreads();

  // =========================== P.V ===============================================
  float oacc[GT][NIV];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
#pragma unroll
    for (int i = 0; i < NIV; ++i) oacc[g][i] = 0.f;
  }

  auto pv_block = [&](int blk, uint4 (&vv)[NIV]) {
    if (blk == num_seq_blocks - 1) {   // tokens >= seq_len of the last block: garbage (0 * NaN must not reach O)
#pragma unroll
      for (int i = 0; i < NIV; ++i) vv[i] = mask_tail<CT>(vv[i], blk * BS + tq * X, seq_len);
    }
    uint4 pr[GT][QP];
#pragma unroll
    for (int g = 0; g < GT; ++g) {
#pragma unroll
      for (int p = 0; p < QP; ++p) {
        pr[g][p] = *reinterpret_cast<const uint4*>(probs + g * logits_cap + (blk - start_block) * BS +
                                                   tq * X + p * XT);
      }
    }
#pragma unroll
    for (int i = 0; i < NIV; ++i) {
      uint4 vt[QP];
      Piece<T, CT>::cvt(vv[i], vt);
#pragma unroll
      for (int g = 0; g < GT; ++g) oacc[g][i] = Piece<T, CT>::dot(vt, pr[g], oacc[g][i]);
    }
  };

  if constexpr (HS != 0) {
    // two buffers: the load of item i + 2 goes out as soon as item i has been multiplied
    for (int i = 0; i < nitems; i += 2) {
      pv_block(start_block + item_block(i), vpre0);
      if (i + 2 < nitems) load_v(start_block + item_block(i + 2), vpre0);
      if (i + 1 < nitems) {
        pv_block(start_block + item_block(i + 1), vpre1);
        if (i + 3 < nitems) load_v(start_block + item_block(i + 3), vpre1);
      }
    }
  } else {
    for (int blk = start_block + wave; blk < end_block; blk += nwaves) {
      uint4 v0[NIV];
      load_v(blk, v0);
      pv_block(blk, v0);
    }
  }

  // lanes that hold the same V row (different token pieces) -> lane with tq == 0
#pragma unroll
  for (int g = 0; g < GT; ++g) {
#pragma unroll
    for (int i = 0; i < NIV; ++i) {
#pragma unroll
      for (int m = 1; m < TPP; m <<= 1) oacc[g][i] += __shfl_xor(oacc[g][i], m, 64);
    }
  }

  // cross-wave sum through LDS (reuses the logits region)
  __syncthreads();
  float* osm = logits;  // [nwaves][GT][D]
  if (tq == 0) {
#pragma unroll
    for (int g = 0; g < GT; ++g) {
#pragma unroll
      for (int i = 0; i < NIV; ++i) {
        const int d = dsub + DPI * i;
        if (d < D) osm[(wave * GT + g) * D + d] = oacc[g][i];
      }
    }
  }
  __syncthreads();
  float vs = 1.f;
  if constexpr (KV8) vs = *v_scale;
  for (int i = tid; i < nheads * D; i += nthreads) {
    const int g = i / D;
    const int d = i - g * D;
    float acc = 0.f;
    for (int w = 0; w < nwaves; ++w) acc += osm[(w * GT + g) * D + d];
    const int64_t o = (((int64_t)seq * num_heads + head0 + g) * num_parts + part) * D + d;
    out[o] = from_f32<T>(KV8 ? acc * vs : acc);
  }
}

// ---------------------------------------------------------------------------
// v2 reduce: grid (num_heads, num_seqs), 128 threads.
// ref: csrc/attention/attention_kernels.cuh:553-658.
template <typename T>
__global__ __launch_bounds__(128) void paged_attention_reduce_kernel(
    T* __restrict__ out, const float* __restrict__ exp_sums,
    const float* __restrict__ max_logits, const T* __restrict__ tmp_out,
    const int* __restrict__ seq_lens, int head_size, int max_num_partitions,
    int partition_size) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_w = reinterpret_cast<float*>(smem);  // [max_num_partitions]
  __shared__ float red[16];
  __shared__ float s_inv;
  const int head = blockIdx.x;
  const int num_heads = gridDim.x;
  const int seq = blockIdx.y;
  const int seq_len = seq_lens[seq];
  const int np = (seq_len + partition_size - 1) / partition_size;
  const int64_t base = ((int64_t)seq * num_heads + head) * max_num_partitions;
  T* o = out + ((int64_t)seq * num_heads + head) * head_size;
  const T* tmp = tmp_out + base * head_size;
  if (np == 1) {
    for (int i = threadIdx.x; i < head_size; i += blockDim.x) o[i] = tmp[i];
    return;
  }
  float m = -3.402823466e+38f;
  for (int j = threadIdx.x; j < np; j += blockDim.x) m = fmaxf(m, max_logits[base + j]);
  m = block_reduce<true>(m, red);
  float s = 0.f;
  for (int j = threadIdx.x; j < np; j += blockDim.x) {
    const float w = exp_sums[base + j] * expf(max_logits[base + j] - m);
    s_w[j] = w;
    s += w;
  }
  s = block_reduce<false>(s, red);
  if (threadIdx.x == 0) s_inv = __fdividef(1.0f, s + 1e-6f);
  __syncthreads();
  const float inv = s_inv;
  for (int i = threadIdx.x; i < head_size; i += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < np; ++j) acc += to_f32(tmp[(int64_t)j * head_size + i]) * s_w[j] * inv;
    o[i] = from_f32<T>(acc);
  }
}

// v2 reduce + dynamic per-token fp8 quantisation of the attention output (the input of an fp8 o_proj) in one launch:
// one workgroup per sequence, wave h replays paged_attention_reduce_kernel for head h (np <= 64 partitions: the
// weights of a head live in one wave, the same wave_max / wave_sum butterflies; that kernel's block_reduce adds an
// exact 0 from its second wave), then the row maximum over all heads and the bytes of
// dynamic_per_token_scaled_fp8_quant (fp8_quant.hip: s = max(absmax / 448, 1 / (448 * 512)), q = sat(float(o) / s)).
// Bit-identical to the two launches (tests/test_gpu_paged_attention.py).
template <typename T>
__global__ __launch_bounds__(1024) void paged_attention_reduce_quant_kernel(
    uint8_t* __restrict__ out_q, float* __restrict__ out_scales, const float* __restrict__ exp_sums,
    const float* __restrict__ max_logits, const T* __restrict__ tmp_out, const int* __restrict__ seq_lens,
    int num_heads, int head_size, int max_num_partitions, int partition_size) {
  __shared__ float red[16];
  __shared__ float s_scale;
  const int lane = threadIdx.x & 63;
  const int head = threadIdx.x >> 6;          // blockDim.x == 64 * num_heads
  const int seq = blockIdx.x;
  const int seq_len = seq_lens[seq];
  const int np = (seq_len + partition_size - 1) / partition_size;
  const int64_t base = ((int64_t)seq * num_heads + head) * max_num_partitions;
  const T* tmp = tmp_out + base * head_size;
  constexpr int kMaxD = 4;                    // head_size <= 256: up to 4 elements per lane
  T o[kMaxD];
  float amax = 0.f;
  float w = 0.f, inv = 1.f;
  if (np > 1) {
    float m = lane < np ? max_logits[base + lane] : -3.402823466e+38f;
    m = wave_max(m);
    w = lane < np ? exp_sums[base + lane] * expf(max_logits[base + lane] - m) : 0.f;
    float sum = wave_sum(w);
    sum = sum + 0.f;                          // (the second wave of paged_attention_reduce_kernel's block_reduce)
    inv = __fdividef(1.0f, sum + 1e-6f);
  }
#pragma unroll
  for (int c = 0; c < kMaxD; ++c) {
    const int i = lane + 64 * c;
    o[c] = from_f32<T>(0.f);
    if (i < head_size) {
      if (np == 1) {
        o[c] = tmp[i];
      } else {
        float acc = 0.f;
        for (int j = 0; j < np; ++j) acc += to_f32(tmp[(int64_t)j * head_size + i]) * __shfl(w, j, 64) * inv;
        o[c] = from_f32<T>(acc);
      }
      amax = fmaxf(amax, fabsf(to_f32(o[c])));
    }
  }
  amax = block_reduce<true>(amax, red);
  if (threadIdx.x == 0) {
    const float sc = fmaxf(amax / kFp8Max, kFp8MinScale);
    out_scales[seq] = sc;
    s_scale = sc;
  }
  __syncthreads();
  const RowDiv rdiv = make_row_div(s_scale);
  uint8_t* q = out_q + ((int64_t)seq * num_heads + head) * head_size;
#pragma unroll
  for (int c = 0; c < kMaxD; ++c) {
    const int i = lane + 64 * c;
    if (i < head_size) q[i] = f32_to_fp8_sat(row_div(to_f32(o[c]), rdiv));
  }
}

struct PaArgs {
  void* out;
  float* exp_sums;
  float* max_logits;
  void* tmp_out;
  const void* query;
  const void* key_cache;
  const void* value_cache;
  int num_seqs, num_heads, num_kv_heads, head_size, block_size;
  float scale;
  const int* block_tables;
  const int* seq_lens;
  int max_num_blocks_per_seq, max_seq_len;
  const float* alibi_slopes;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  int partition_size;  // 0 => v1
  hipStream_t stream;
  int kv_cache_dtype = MI355X_KV_AUTO;
  const float* k_scale = nullptr;
  const float* v_scale = nullptr;
  const FusedQkv* fused = nullptr;   // mi355x_paged_attention_fused_qkv
};

// Launch geometry shared by the launcher and mi355x_paged_attention_v1_max_seq_len.
struct PaPlan {
  int gt, tiles, threads, logits_cap, num_parts;
  size_t smem;
};
constexpr size_t kPaLdsLimit = 160 * 1024;

static PaPlan pa_plan(int num_seqs, int num_heads, int num_kv_heads, int head_size, int block_size,
                      int elt_size, int max_seq_len, int partition_size) {
  PaPlan p;
  const int q_per_kv = num_heads / num_kv_heads;
  p.gt = q_per_kv >= 5 ? 8 : (q_per_kv >= 3 ? 4 : q_per_kv);  // 1, 2, 4 or 8 heads per workgroup
  p.tiles = (q_per_kv + p.gt - 1) / p.gt;
  const int padded_len = ((max_seq_len + block_size - 1) / block_size) * block_size;
  p.logits_cap = partition_size > 0 ? partition_size : padded_len;
  p.num_parts = partition_size > 0 ? (max_seq_len + partition_size - 1) / partition_size : 1;
  // Few workgroups (v1: one per (seq, kv head)) leave half of a CU's wave slots empty with
  // 4-wave workgroups; 8 waves per workgroup put the same number of loads in flight per CU as the
  // partitioned launch does (measured at 64 seqs x 8 kv heads, ctx 1088: see DESIGN.md §3).
  const int64_t wgs = (int64_t)num_kv_heads * p.tiles * num_seqs * p.num_parts;
  p.threads = wgs < 768 ? kPaMaxThreads : kPaThreads;
  // the cross-wave output buffer [waves][GT][D] aliases the logits region
  const int min_cap = (p.threads / 64) * head_size;
  if (p.logits_cap < min_cap) p.logits_cap = min_cap;
  p.logits_cap = (p.logits_cap + 63) & ~63;
  // + the block ids of the (sequence, partition): logits_cap / block_size ints
  p.smem = pa_scratch_bytes(p.gt) + (size_t)p.gt * 256 * elt_size + (size_t)p.gt * p.logits_cap * (4 + elt_size) +
           (size_t)(p.logits_cap / block_size) * 4;
  return p;
}

template <typename T, typename CT, int BS, int GT, int HS, bool FQ = false>
static int launch_pa_inst(const PaArgs& a, int tiles, int num_parts, int logits_cap,
                          size_t smem, int threads) {
  auto kern = paged_attention_kernel<T, CT, BS, GT, HS, FQ>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      set_error("paged_attention: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
      return MI355X_EUNSUPPORTED;
    }
  }
  dim3 grid(a.num_kv_heads * tiles, a.num_seqs, num_parts);
  T* dst = static_cast<T*>(a.partition_size > 0 ? a.tmp_out : a.out);
  hipLaunchKernelGGL(kern, grid, dim3(threads), smem, a.stream, a.exp_sums, a.max_logits, dst,
                     static_cast<const T*>(a.query), static_cast<const CT*>(a.key_cache),
                     static_cast<const CT*>(a.value_cache), a.num_heads, a.num_kv_heads,
                     a.head_size, a.scale, a.block_tables, a.seq_lens,
                     a.max_num_blocks_per_seq, a.alibi_slopes, a.q_stride, a.kv_block_stride,
                     a.kv_head_stride, a.partition_size, logits_cap, a.k_scale, a.v_scale,
                     FQ ? *a.fused : FusedQkv{});
  return check_launch("paged_attention");
}

template <typename T, typename CT, int BS>
static int launch_pa_bs(const PaArgs& a) {
  const PaPlan p = pa_plan(a.num_seqs, a.num_heads, a.num_kv_heads, a.head_size, BS, (int)sizeof(T),
                           a.max_seq_len, a.partition_size);
  MI355X_REQUIRE(p.smem <= kPaLdsLimit, MI355X_EUNSUPPORTED,
                 "paged_attention_v1: max_seq_len %d needs %zu B of LDS (> 160 KiB); use "
                 "paged_attention_v2 (mi355x_paged_attention_v1_max_seq_len gives the limit)",
                 a.max_seq_len, p.smem);
  const bool fast = (a.head_size == 128) && (BS == 16) && (sizeof(T) == 2);
  if (a.fused != nullptr) {
    // fused qkv prologue: the shapes of the decode fast path with ONE workgroup per (sequence, kv head)
    if constexpr (BS == 16 && sizeof(T) == 2 && std::is_same<T, CT>::value) {
      if (fast && p.gt == 4 && p.tiles == 1 && a.num_heads / a.num_kv_heads == 4)
        return launch_pa_inst<T, CT, BS, 4, 128, true>(a, p.tiles, p.num_parts, p.logits_cap, p.smem, p.threads);
      if (fast && p.gt == 8 && p.tiles == 1 && a.num_heads / a.num_kv_heads == 8)
        return launch_pa_inst<T, CT, BS, 8, 128, true>(a, p.tiles, p.num_parts, p.logits_cap, p.smem, p.threads);
    }
    return 1;   // not applicable: the caller runs qkv_rope_cache + paged_attention
  }
#define PA_CASE(GTV)                                                                            \
  if (p.gt == GTV) {                                                                            \
    if (fast) {                                                                                 \
      if constexpr (BS == 16 && sizeof(T) == 2)                                                 \
        return launch_pa_inst<T, CT, BS, GTV, 128>(a, p.tiles, p.num_parts, p.logits_cap, p.smem, \
                                                   p.threads);                                  \
    }                                                                                           \
    return launch_pa_inst<T, CT, BS, GTV, 0>(a, p.tiles, p.num_parts, p.logits_cap, p.smem,      \
                                             p.threads);                                        \
  }
  PA_CASE(1)
  PA_CASE(2)
  PA_CASE(4)
  PA_CASE(8)
#undef PA_CASE
  set_error("paged_attention: internal dispatch error");
  return MI355X_EINVAL;
}

template <typename T>
static int launch_pa(const PaArgs& a) {
  if (a.kv_cache_dtype == MI355X_KV_FP8_E5M2) {
    switch (a.block_size) {
      case 16: return launch_pa_bs<T, e5m2_t, 16>(a);
      case 32: return launch_pa_bs<T, e5m2_t, 32>(a);
      default:
        set_error("Unsupported block size with an fp8 KV cache: %d (16 or 32)", a.block_size);
        return MI355X_EUNSUPPORTED;
    }
  }
  if (a.kv_cache_dtype == MI355X_KV_FP8_E4M3) {
    switch (a.block_size) {   // a 16-byte piece of the byte cache is 16 tokens of a V row
      case 16: return launch_pa_bs<T, uint8_t, 16>(a);
      case 32: return launch_pa_bs<T, uint8_t, 32>(a);
      default:
        set_error("Unsupported block size with an fp8 KV cache: %d (16 or 32)", a.block_size);
        return MI355X_EUNSUPPORTED;
    }
  }
  switch (a.block_size) {
    case 8: return launch_pa_bs<T, T, 8>(a);
    case 16: return launch_pa_bs<T, T, 16>(a);
    case 32: return launch_pa_bs<T, T, 32>(a);
    default:
      set_error("Unsupported block size: %d", a.block_size);
      return MI355X_EUNSUPPORTED;
  }
}

static int validate_pa(const PaArgs& a, const char* name) {
  MI355X_REQUIRE(a.num_seqs >= 0 && a.num_heads > 0 && a.num_kv_heads > 0 && a.head_size > 0 &&
                     a.max_num_blocks_per_seq >= 0 && a.max_seq_len >= 0,
                 MI355X_EINVAL, "%s: bad sizes", name);
  MI355X_REQUIRE(a.num_heads % a.num_kv_heads == 0, MI355X_EINVAL,
                 "%s: num_heads %d not a multiple of num_kv_heads %d", name, a.num_heads,
                 a.num_kv_heads);
  switch (a.head_size) {  // ref: paged_attention_v1.cu:90-124
    case 32: case 64: case 80: case 96: case 112: case 120: case 128: case 192: case 256: break;
    default:
      set_error("Unsupported head size: %d", a.head_size);
      return MI355X_EUNSUPPORTED;
  }
  if (a.num_seqs == 0) return MI355X_OK;
  MI355X_REQUIRE(a.out && a.query && a.key_cache && a.value_cache && a.block_tables && a.seq_lens,
                 MI355X_EINVAL, "%s: null pointer", name);
  MI355X_REQUIRE(a.num_seqs <= 65535, MI355X_EUNSUPPORTED, "%s: num_seqs %d > 65535", name,
                 a.num_seqs);
  // ref: csrc/quantization/fp8/metax/quant_utils.cuh:29-42 rejects everything but "auto"; the
  // e4m3 cache is this build's SURVEY §8f-3 row (upstream vLLM's "fp8" / "fp8_e4m3")
  MI355X_REQUIRE(a.kv_cache_dtype == MI355X_KV_AUTO || a.kv_cache_dtype == MI355X_KV_FP8_E4M3 ||
                     a.kv_cache_dtype == MI355X_KV_FP8_E5M2,
                 MI355X_EUNSUPPORTED, "Unsupported data type of kv cache: id %d", a.kv_cache_dtype);
  if (a.kv_cache_dtype != MI355X_KV_AUTO) {
    MI355X_REQUIRE(a.k_scale && a.v_scale, MI355X_EINVAL, "%s: fp8 KV cache needs k_scale / v_scale", name);
    MI355X_REQUIRE(a.head_size % 16 == 0, MI355X_EUNSUPPORTED,
                   "%s: fp8 KV cache needs head_size %% 16 == 0 (got %d)", name, a.head_size);
  }
  return MI355X_OK;
}

}  // namespace mi355x

using namespace mi355x;

extern "C" {

// returns 1 (no error set) when the fused form does not apply to the shapes: the caller then runs
// mi355x_qkv_rope_cache followed by mi355x_paged_attention_v1 / _v2
static int fused_qkv_impl(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const float* slabs, int sk, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int x, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, int64_t kv_block_stride,
    int64_t kv_head_stride, int partition_size, int dtype, mi355x_stream stream,
    const float* a_scales, int a_scales_numel, const float* b_scales, int b_scales_numel,
    void* out_q, float* out_scales, const char* name) {
  MI355X_REQUIRE(sk >= 0 && partition_size >= 0, MI355X_EINVAL, "%s: bad sizes", name);
  if (dtype != MI355X_BF16 && dtype != MI355X_F16) return 1;
  if (head_size != 128 || block_size != 16 || x != 8) return 1;
  // quantised output: the reduce + quant kernel of the partitioned form (one wave per head, <= 64 partitions)
  const bool quant = out_q != nullptr;
  if (quant && (partition_size == 0 || num_heads > 16 ||
                (max_seq_len + partition_size - 1) / partition_size > 64))
    return 1;
  PaArgs a{};
  a.out = out; a.exp_sums = exp_sums; a.max_logits = max_logits; a.tmp_out = tmp_out;
  a.query = qkv;   // (unused by the fused kernel; non-null for validate_pa)
  a.key_cache = key_cache; a.value_cache = value_cache;
  a.num_seqs = num_seqs; a.num_heads = num_heads; a.num_kv_heads = num_kv_heads;
  a.head_size = head_size; a.block_size = block_size; a.scale = scale;
  a.block_tables = block_tables; a.seq_lens = seq_lens;
  a.max_num_blocks_per_seq = max_num_blocks_per_seq; a.max_seq_len = max_seq_len;
  a.alibi_slopes = nullptr; a.q_stride = qkv_stride;
  a.kv_block_stride = kv_block_stride; a.kv_head_stride = kv_head_stride;
  a.partition_size = partition_size; a.stream = static_cast<hipStream_t>(stream);
  if (quant) a.out = out_q;   // (validate_pa wants a non-null output)
  int rc = validate_pa(a, name);
  if (rc != MI355X_OK || num_seqs == 0) return rc;
  // positions / cos_sin_cache both NULL: no rotary (q and k already rotated; the cache write is all that is folded in)
  MI355X_REQUIRE(slot_mapping && (sk == 0 || slabs) && ((positions == nullptr) == (cos_sin_cache == nullptr)),
                 MI355X_EINVAL, "%s: null pointer", name);
  MI355X_REQUIRE(partition_size == 0 || (exp_sums && max_logits && tmp_out), MI355X_EINVAL,
                 "%s: the partitioned form needs exp_sums / max_logits / tmp_out", name);
  MI355X_REQUIRE(partition_size % 16 == 0, MI355X_EINVAL,
                 "%s: partition_size must be 0 or a multiple of the block size (16)", name);
  MI355X_REQUIRE(((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(slabs) |
                   reinterpret_cast<uintptr_t>(cos_sin_cache) | reinterpret_cast<uintptr_t>(key_cache)) & 15) == 0 &&
                     qkv_stride % 8 == 0,
                 MI355X_EINVAL, "%s: 16-byte aligned qkv / slabs / cos_sin / cache", name);
  const int width = (num_heads + 2 * num_kv_heads) * head_size;
  MI355X_REQUIRE(b_scales == nullptr ||
                     (a_scales && sk > 0 && (a_scales_numel == 1 || a_scales_numel == num_seqs) &&
                      (b_scales_numel == 1 || b_scales_numel == width)),
                 MI355X_EINVAL, "%s: scales need slabs (sk > 0), per-tensor or per-token / per-column", name);
  MI355X_REQUIRE(!quant || out_scales, MI355X_EINVAL, "%s: quantised output needs out_scales", name);
  FusedQkv f{qkv, qkv_stride, sk > 0 ? slabs : nullptr, sk, (int64_t)num_seqs * width, positions, cos_sin_cache,
             slot_mapping, SlabScales{a_scales, b_scales, a_scales_numel > 1, b_scales_numel > 1}};
  a.fused = &f;
  rc = dtype == MI355X_BF16 ? launch_pa<bf16_t>(a) : launch_pa<f16_t>(a);
  if (rc != MI355X_OK || partition_size == 0) return rc;
  // partitioned: the same reduce as paged_attention_v2 (or reduce + per-token fp8 quantisation)
  const int max_parts = (max_seq_len + partition_size - 1) / partition_size;
  return MI355X_DISPATCH_HALF(dtype, [&] {
    if (quant) {
      hipLaunchKernelGGL(paged_attention_reduce_quant_kernel<scalar_t>, dim3(num_seqs), dim3(64 * num_heads), 0,
                         a.stream, static_cast<uint8_t*>(out_q), out_scales, exp_sums, max_logits,
                         static_cast<const scalar_t*>(tmp_out), seq_lens, num_heads, head_size, max_parts,
                         partition_size);
      return check_launch("paged_attention_fused_qkv_reduce_quant");
    }
    hipLaunchKernelGGL(paged_attention_reduce_kernel<scalar_t>, dim3(num_heads, num_seqs),
                       dim3(128), (size_t)(max_parts > 0 ? max_parts : 1) * sizeof(float), a.stream,
                       static_cast<scalar_t*>(out), exp_sums, max_logits,
                       static_cast<const scalar_t*>(tmp_out), seq_lens, head_size, max_parts,
                       partition_size);
    return check_launch("paged_attention_fused_qkv_reduce");
  });
}

int mi355x_paged_attention_fused_qkv(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const float* slabs, int sk, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int x, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, int64_t kv_block_stride,
    int64_t kv_head_stride, int partition_size, int dtype, mi355x_stream stream) {
  return fused_qkv_impl(out, exp_sums, max_logits, tmp_out, qkv, qkv_stride, slabs, sk, positions, cos_sin_cache,
                        slot_mapping, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size,
                        block_size, x, scale, block_tables, seq_lens, max_num_blocks_per_seq, max_seq_len,
                        kv_block_stride, kv_head_stride, partition_size, dtype, stream, nullptr, 0, nullptr, 0,
                        nullptr, nullptr, "paged_attention_fused_qkv");
}

int mi355x_paged_attention_fused_qkv_w8(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const float* slabs, int sk, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int x, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, int64_t kv_block_stride,
    int64_t kv_head_stride, int partition_size, int dtype, const float* a_scales, int a_scales_numel,
    const float* b_scales, int b_scales_numel, void* out_q, float* out_scales, mi355x_stream stream) {
  return fused_qkv_impl(out, exp_sums, max_logits, tmp_out, qkv, qkv_stride, slabs, sk, positions, cos_sin_cache,
                        slot_mapping, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size,
                        block_size, x, scale, block_tables, seq_lens, max_num_blocks_per_seq, max_seq_len,
                        kv_block_stride, kv_head_stride, partition_size, dtype, stream, a_scales, a_scales_numel,
                        b_scales, b_scales_numel, out_q, out_scales, "paged_attention_fused_qkv_w8");
}

int mi355x_paged_attention_v1_max_seq_len(int num_seqs, int num_heads, int num_kv_heads,
                                          int head_size, int block_size, int dtype) {
  MI355X_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0 && head_size > 0 &&
                     (block_size == 8 || block_size == 16 || block_size == 32),
                 MI355X_EINVAL, "paged_attention_v1_max_seq_len: bad sizes");
  MI355X_REQUIRE(dtype == MI355X_F16 || dtype == MI355X_BF16 || dtype == MI355X_F32, MI355X_EUNSUPPORTED,
                 "unsupported dtype id %d", dtype);
  const int esz = dtype_size(dtype);
  const PaPlan p = pa_plan(num_seqs > 0 ? num_seqs : 1, num_heads, num_kv_heads, head_size, block_size,
                           esz, block_size, 0);
  const size_t fixed = pa_scratch_bytes(p.gt) + (size_t)p.gt * 256 * esz;
  // per token: gt * (fp32 logit + scalar_t probability) + 4 / block_size bytes of block id
  int64_t cap = (int64_t)((kPaLdsLimit - fixed) * block_size / ((size_t)p.gt * (4 + esz) * block_size + 4));
  cap &= ~(int64_t)63;                       // the launcher rounds the logits capacity up to 64
  return (int)cap;                           // = the largest max_seq_len whose padded length fits
}

int mi355x_paged_attention_v1(void* out, const void* query, const void* key_cache,
                              const void* value_cache, int num_seqs, int num_heads,
                              int num_kv_heads, int head_size, int block_size,
                              float scale, const int* block_tables,
                              const int* seq_lens, int max_num_blocks_per_seq,
                              int max_seq_len, const float* alibi_slopes,
                              int64_t q_stride, int64_t kv_block_stride,
                              int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                              const float* k_scale, const float* v_scale, mi355x_stream stream) {
  PaArgs a{out, nullptr, nullptr, nullptr, query, key_cache, value_cache, num_seqs, num_heads,
           num_kv_heads, head_size, block_size, scale, block_tables, seq_lens,
           max_num_blocks_per_seq, max_seq_len, alibi_slopes, q_stride, kv_block_stride,
           kv_head_stride, 0, static_cast<hipStream_t>(stream), kv_cache_dtype, k_scale, v_scale};
  int rc = validate_pa(a, "paged_attention_v1");
  if (rc || num_seqs == 0) return rc;
  return MI355X_DISPATCH_FLOAT(dtype, [&] { return launch_pa<scalar_t>(a); });
}

int mi355x_paged_attention_v2_ps(void* out, float* exp_sums, float* max_logits,
                                 void* tmp_out, const void* query, const void* key_cache,
                                 const void* value_cache, int num_seqs, int num_heads,
                                 int num_kv_heads, int head_size, int block_size,
                                 float scale, const int* block_tables,
                                 const int* seq_lens, int max_num_blocks_per_seq,
                                 int max_seq_len, const float* alibi_slopes,
                                 int64_t q_stride, int64_t kv_block_stride,
                                 int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                                 const float* k_scale, const float* v_scale, int partition_size,
                                 mi355x_stream stream) {
  MI355X_REQUIRE(partition_size > 0 && block_size > 0 && partition_size % block_size == 0 && partition_size % 16 == 0,
                 MI355X_EINVAL, "paged_attention_v2: partition_size %d must be a positive multiple of the block size "
                 "and of 16", partition_size);
  PaArgs a{out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
           num_heads, num_kv_heads, head_size, block_size, scale, block_tables, seq_lens,
           max_num_blocks_per_seq, max_seq_len, alibi_slopes, q_stride, kv_block_stride,
           kv_head_stride, partition_size, static_cast<hipStream_t>(stream),
           kv_cache_dtype, k_scale, v_scale};
  int rc = validate_pa(a, "paged_attention_v2");
  if (rc || num_seqs == 0) return rc;
  MI355X_REQUIRE(exp_sums && max_logits && tmp_out, MI355X_EINVAL,
                 "paged_attention_v2: null workspace pointer");
  rc = MI355X_DISPATCH_FLOAT(dtype, [&] { return launch_pa<scalar_t>(a); });
  if (rc) return rc;
  const int max_parts = (max_seq_len + partition_size - 1) / partition_size;
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    hipLaunchKernelGGL(paged_attention_reduce_kernel<scalar_t>, dim3(num_heads, num_seqs),
                       dim3(128), (size_t)(max_parts > 0 ? max_parts : 1) * sizeof(float), a.stream,
                       static_cast<scalar_t*>(out), exp_sums, max_logits,
                       static_cast<const scalar_t*>(tmp_out), seq_lens, head_size, max_parts,
                       partition_size);
    return check_launch("paged_attention_v2_reduce");
  });
}

int mi355x_paged_attention_v2(void* out, float* exp_sums, float* max_logits,
                              void* tmp_out, const void* query, const void* key_cache,
                              const void* value_cache, int num_seqs, int num_heads,
                              int num_kv_heads, int head_size, int block_size,
                              float scale, const int* block_tables,
                              const int* seq_lens, int max_num_blocks_per_seq,
                              int max_seq_len, const float* alibi_slopes,
                              int64_t q_stride, int64_t kv_block_stride,
                              int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                              const float* k_scale, const float* v_scale, mi355x_stream stream) {
  return mi355x_paged_attention_v2_ps(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
                                      num_heads, num_kv_heads, head_size, block_size, scale, block_tables, seq_lens,
                                      max_num_blocks_per_seq, max_seq_len, alibi_slopes, q_stride, kv_block_stride,
                                      kv_head_stride, dtype, kv_cache_dtype, k_scale, v_scale,
                                      MI355X_PA_PARTITION_SIZE, stream);
}

}  // extern "C"
