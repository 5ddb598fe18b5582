// fp8_gemm.hip — FP8 (OCP e4m3fn) weight/activation GEMM on gfx950 MFMA, behind the
// reference's cutlass_scaled_mm schema (csrc/torch_bindings.cpp:251-256).
//
// The reference implements that schema for int8 only (scaled_mm_entry.cu:34-39 ->
// scaled_mm_c2x.cu) and answers "fp8 unsupported" (scaled_mm_entry.cu:22-24,
// platform.py:412-414); this is the new capability north_star asks for.  Semantics follow
// the reference's test-side definition tests/kernels/utils.py:1231-1270 (baseline_scaled_mm):
//     out[M,N] = (a_scales . A[M,K]) x (b_scales . B[K,N]) (+ bias[N]),  fp32 accumulate,
// A row-major e4m3fn, B COLUMN-major e4m3fn (b.stride(0) == 1), scales per-tensor or
// per-row (A) / per-column (B), out bf16 / f16.
//
// MI355X design: v_mfma_f32_16x16x32_fp8_fp8 (each lane feeds 8 consecutive-k bytes of A
// and of B).  A K-tile is 64 bytes deep: lane group lr owns bytes 16*lr .. 16*lr+15 of the
// tile for BOTH operands (low 8 bytes -> first MFMA, high 8 bytes -> second), so one
// 16-B load per lane feeds two MFMAs and no k-permutation is ever materialised.
//   large M : 128 x 256 tile / 256 threads; each wave owns 64 columns, B straight from
//             global to VGPRs, A staged through LDS in fragment-major order (lane-linear
//             ds_read_b128), issue-early / write-late double buffering, XCD-aware tiles.
//   small M : 64 x 64 tile, the 4 waves split K and are summed through LDS; optional
//             split-K across workgroups through fp32 partial slabs [sk][m][n] summed in slab order by a finish
//             kernel.  (Rounds 1-2 zeroed ONE [m, n] workspace with hipMemsetAsync and added into it with atomics:
//             inside a captured HIP graph that sequence returned garbage from the second replay on — the cause of
//             the non-finite logits of the `--quant fp8` job, scripts/debug_fp8_graph.py)
//
// int8 (W8A8): the SAME kernels instantiated with v_mfma_i32_16x16x32_i8 —
// identical operand shape (8 consecutive-k bytes per lane), exact int32 accumulation (split-K
// through int32 partial slabs), epilogue out = T(a_s * (b_s * float(acc)) + bias) in fp32 — the
// reference's int8 cutlass_scaled_mm (csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:34-39,
// :84-140; epilogue order as tests/kernels/utils.py baseline_scaled_mm).
#include <cstdio>
#include <cstdlib>

#include "w4a16.cuh"   // pack_a_kernel / frag_swz: the packed operand image of the prefill path

namespace mi355x {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// 16 bytes that no other wave will read again: do not keep the line in L2 (global_load_dwordx4 ... nt)
__device__ __forceinline__ uint4 load_nt16(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

struct OpFp8 {
  static constexpr bool kWide = true;    // has a 16x16x128 MFMA (run32)
  typedef f32x4_t acc_t;
  typedef float elem_t;
  static __device__ __forceinline__ acc_t run(uint64_t a, uint64_t b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
  }
  // one 64-k piece (16 B per lane on each side): two 16x16x32 MFMAs
  static __device__ __forceinline__ acc_t run16(uint4 a, uint4 b, acc_t c) {
    c = run(((uint64_t)a.y << 32) | a.x, ((uint64_t)b.y << 32) | b.x, c);
    return run(((uint64_t)a.w << 32) | a.z, ((uint64_t)b.w << 32) | b.z, c);
  }
  // two 64-k pieces in one 16x16x128 MFMA (gfx950 f8f6f4 form: cbsz = blgp = 0 selects e4m3 on
  // both sides, the E8M0 block scales are 127 = 2^0)
  static __device__ __forceinline__ acc_t run32(uint4 a0, uint4 a1, uint4 b0, uint4 b1, acc_t c) {
    const i32x8_t a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w,
                       (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const i32x8_t b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w,
                       (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  }
  // (a_s . A)(b_s . B) + bias, the order the fp8 oracle uses
  static __device__ __forceinline__ float finish(float acc, float as, float bs, float bi) {
    return scaled_finish_fp8(acc, as, bs, bi);   // (common.cuh: the slab consumers of the deferred form use it too)
  }
};
struct OpI8 {
  // (int8 through the two-slice stage with both operands in place — two 16x16x64 per operand pair — was measured
  //  SLOWER than its 4-stage ping-pong ring on images: 383 -> 415 us per call at M = 8192, 78 -> 94 at M = 576,
  //  profiles/r03_fp8_operands_in_place.txt)
  static constexpr bool kWide = false;
  typedef i32x4_t acc_t;
  typedef int elem_t;
  static __device__ __forceinline__ acc_t run(uint64_t a, uint64_t b, acc_t c) {
    return __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a, (long)b, c, 0, 0, 0);
  }
  // one 64-k piece in one 16x16x64 MFMA
  static __device__ __forceinline__ acc_t run16(uint4 a, uint4 b, acc_t c) {
    const i32x4_t av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w};
    const i32x4_t bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
  }
  static __device__ __forceinline__ acc_t run32(uint4 a0, uint4 a1, uint4 b0, uint4 b1, acc_t c) {
    return run16(a1, b1, run16(a0, b0, c));
  }
  static __device__ __forceinline__ float finish(int acc, float as, float bs, float bi) {
#pragma clang fp contract(off)
    const float v = as * (bs * (float)acc);
    return v + bi;
  }
};

template <typename T>
__device__ __forceinline__ T out_cast(float v) {
  return from_f32<T>(v);
}

struct Fp8Args {
  void* out;
  const uint8_t* a;
  const uint8_t* b;
  const float* a_scales;
  int a_scales_numel;
  const float* b_scales;
  int b_scales_numel;
  const void* bias;
  int m, n, k;
  int64_t lda, ldb, ldc;
  float* ws;
  int64_t ws_elems;
  hipStream_t stream;
  int* defer_sk = nullptr;   // mi355x_scaled_mm_fp8_deferred: leave a K split's slabs to the consumer, report sk here
  const void* b_image = nullptr;   // mi355x_scaled_mm_prepack: the weights' operand image (rows interleaved by 4),
                                   // built once at load time; `b` is then unused by the packed path
};

// ------------------------------------------------------------------------- large M
constexpr int kF8BM = 128;
constexpr int kF8BN = 256;
constexpr int kF8BK = 64;  // bytes of K per tile

template <typename T, typename Op>
__global__ __launch_bounds__(256, 2) void fp8_gemm_large_kernel(
    T* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, int a_per_row, const float* __restrict__ b_scales,
    int b_per_col, const T* __restrict__ bias, int m, int n, int k, int64_t lda, int64_t ldb,
    int64_t ldc, int num_m_blocks, int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* a_lds = reinterpret_cast<uint4*>(smem);  // [2][8 pieces][64 lanes]

  int tile;
  {
    const int bid = blockIdx.x;
    const int q = num_tiles / 8, r = num_tiles % 8;
    const int xcd = bid % 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int nb = tile / num_m_blocks;
  const int mb = tile - nb * num_m_blocks;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int n0 = nb * kF8BN + wave * 64;
  const bool wave_active = n0 < n;
  const int ktiles = k / kF8BK;

  // A staging: wave w stages pieces (M-tiles) 2w, 2w+1
  const uint8_t* a_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = mb * kF8BM + (wave * 2 + i) * 16 + lc;
    row = row < m ? row : m - 1;
    a_src[i] = a + (int64_t)row * lda + 16 * lr;
  }
  // B rows (columns of the logical B) of this lane for its 4 N-tiles
  const uint8_t* b_src[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int col = n0 + 16 * t + lc;
    col = col < n ? col : n - 1;
    b_src[t] = b + (int64_t)col * ldb + 16 * lr;
  }

  typename Op::acc_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = typename Op::acc_t{0, 0, 0, 0};
  }

  uint4 a_stage[2], b_cur[4], b_nxt[4];
  auto load_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a_stage[i] = *reinterpret_cast<const uint4*>(a_src[i] + kt * kF8BK);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a_lds[(buf * 8 + wave * 2 + i) * 64 + lane] = a_stage[i];
  };
  auto load_b = [&](int kt, uint4 (&dst)[4]) {
    if (wave_active) {
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const uint4*>(b_src[t] + kt * kF8BK);
    }
  };

  load_a(0);
  load_b(0, b_cur);
  store_a(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    const bool has_next = (kt + 1) < ktiles;
    if (has_next) {
      load_a(kt + 1);
      load_b(kt + 1, b_nxt);
    }
    if (wave_active) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint4 af = a_lds[(cur * 8 + i) * 64 + lane];
        const uint64_t a0 = ((uint64_t)af.y << 32) | af.x;
        const uint64_t a1 = ((uint64_t)af.w << 32) | af.z;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint64_t b0 = ((uint64_t)b_cur[t].y << 32) | b_cur[t].x;
          const uint64_t b1 = ((uint64_t)b_cur[t].w << 32) | b_cur[t].z;
          acc[i][t] = Op::run(a0, b0, acc[i][t]);
          acc[i][t] = Op::run(a1, b1, acc[i][t]);
        }
      }
    }
    if (has_next) {
      store_a(cur ^ 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) b_cur[t] = b_nxt[t];
    }
    __syncthreads();
    cur ^= 1;
  }
  if (!wave_active) return;

#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = n0 + 16 * t + lc;
    if (col >= n) continue;
    const float bs = b_scales[b_per_col ? col : 0];
    const float bi = bias ? to_f32(bias[col]) : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * kF8BM + i * 16 + 4 * lr + j;
        if (row < m) {
          const float as = a_scales[a_per_row ? row : 0];
          out[(int64_t)row * ldc + col] = out_cast<T>(Op::finish(acc[i][t][j], as, bs, bi));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- small M
template <typename T, int MT, typename Op>
__global__ __launch_bounds__(256) void fp8_gemm_small_kernel(
    T* __restrict__ out, typename Op::elem_t* __restrict__ ws, const uint8_t* __restrict__ a,
    const uint8_t* __restrict__ b, const float* __restrict__ a_scales, int a_per_row,
    const float* __restrict__ b_scales, int b_per_col, const T* __restrict__ bias, int m, int n,
    int k, int64_t lda, int64_t ldb, int64_t ldc, int ktiles_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Op::elem_t elem_t;
  elem_t* red = reinterpret_cast<elem_t*>(smem);  // [2][MT*16][64]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int total_tiles = k / kF8BK;
  const int kt_begin = blockIdx.y * ktiles_per_split;
  const int kt_end = min(kt_begin + ktiles_per_split, total_tiles);

  const uint8_t* a_src[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int row = i * 16 + lc;
    row = row < m ? row : m - 1;
    a_src[i] = a + (int64_t)row * lda + 16 * lr;
  }
  const uint8_t* b_src[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int col = n0 + 16 * t + lc;
    col = col < n ? col : n - 1;
    b_src[t] = b + (int64_t)col * ldb + 16 * lr;
  }
  typename Op::acc_t acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = typename Op::acc_t{0, 0, 0, 0};
  }
  for (int kt = kt_begin + wave; kt < kt_end; kt += 4) {
    uint4 af[MT], bf[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const uint4*>(a_src[i] + kt * kF8BK);
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const uint4*>(b_src[t] + kt * kF8BK);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const uint64_t a0 = ((uint64_t)af[i].y << 32) | af[i].x;
      const uint64_t a1 = ((uint64_t)af[i].w << 32) | af[i].z;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint64_t b0 = ((uint64_t)bf[t].y << 32) | bf[t].x;
        const uint64_t b1 = ((uint64_t)bf[t].w << 32) | bf[t].z;
        acc[i][t] = Op::run(a0, b0, acc[i][t]);
        acc[i][t] = Op::run(a1, b1, acc[i][t]);
      }
    }
  }
  // cross-wave tree reduction through LDS: element (row, col) at red[row*64 + col]
  auto lds_store = [&](elem_t* dst) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(i * 16 + 4 * lr + j) * 64 + 16 * t + lc] = acc[i][t][j];
  };
  auto lds_add = [&](const elem_t* src) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][t][j] += src[(i * 16 + 4 * lr + j) * 64 + 16 * t + lc];
  };
  constexpr int kSlab = MT * 16 * 64;
  if (wave >= 2) lds_store(red + (wave - 2) * kSlab);
  __syncthreads();
  if (wave < 2) lds_add(red + wave * kSlab);
  __syncthreads();
  if (wave == 1) lds_store(red);
  __syncthreads();
  if (wave != 0) return;
  lds_add(red);

#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = n0 + 16 * t + lc;
    if (col >= n) continue;
    const float bs = b_scales[b_per_col ? col : 0];
    const float bi = bias ? to_f32(bias[col]) : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = i * 16 + 4 * lr + j;
        if (row >= m) continue;
        if (gridDim.y == 1) {
          const float as = a_scales[a_per_row ? row : 0];
          out[(int64_t)row * ldc + col] = out_cast<T>(Op::finish(acc[i][t][j], as, bs, bi));
        } else {
          // split K across workgroups: partial slab [blockIdx.y][m][n], summed in slab order by the finish
          // kernel (no memset node, no atomics: deterministic, and safe to replay from a HIP graph)
          ws[((int64_t)blockIdx.y * m + row) * n + col] = acc[i][t][j];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- decode (M <= 64), streaming
// The weight bytes of an 8-bit layer are read exactly once and nothing has to be dequantised, so the decode GEMM
// is a pure stream: what bounds it is HBM bytes in flight per CU, not issue.  (The round-1/2 small-M kernel above
// — 64 x 64 tile, K waves strided by 4, no prefetch — ran the four Llama-3-70B per-rank projections at 28 us per
// launch, 0.95 TB/s; it stays as the fallback for K % 128 != 0.)
//   * a workgroup = 4 waves = 4 x NT column tiles of 16 and ONE contiguous K range of `steps` x 128 bytes; the
//     activation rows of that range (64 x R bytes, an L2-resident re-read) are copied ONCE by LDS-DMA into LDS
//     (per-lane source addresses, 8 rows x 128 bytes = whole cache lines per copy), one barrier, and from then on
//     the waves run free: no barrier, no shared ring in the loop.
//   * weights go HBM -> VGPR directly (a weight byte is used by exactly one wave; LDS would only add a hop — but in
//     the MFMA's B layout the 4 lanes of a quad sit on 4 different weight rows, 64 requests of 16 bytes per
//     instruction: a timing-only build with quads on 64-byte runs (wrong lanes) ran the 70B rank layer in 51.2
//     instead of 56.4 us, profiles/r03_fp8_decode_whole_line_copies.txt — the bound on what a wave-private LDS ring
//     for the weights could gain, which would have to come out of the activation image's LDS), three
//     register buffers deep: the loads of k-step kk+2 are issued before the MFMAs of kk (8 KiB per wave and k-step
//     at NT = 4, 16 KiB in flight per wave), addresses of the tail clamped instead of branched so that the
//     compiler's wait-count pass keeps counted vmcnt waits.
//   * K split across workgroups (grid.y) through partial slabs [sk][m][n], summed in slab order by the finish
//     kernel together with the scales / bias: deterministic, HIP-graph-replayable.
//   fp8: one v_mfma_scale_f32_16x16x128_f8f6f4 per (row tile, column tile, k-step); int8: two 16x16x64.
template <typename T, typename Op, int MT, int NT, bool NTL = false>
__global__ __launch_bounds__(256) void gemm8_decode_kernel(
    T* __restrict__ out, typename Op::elem_t* __restrict__ ws, const uint8_t* __restrict__ a,
    const uint8_t* __restrict__ b, const float* __restrict__ a_scales, int a_per_row,
    const float* __restrict__ b_scales, int b_per_col, const T* __restrict__ bias, int m, int n, int k,
    int64_t lda, int64_t ldb, int64_t ldc, int steps_per_split) {
  // register buffers of weight k-steps: D - 1 steps (2 KiB x NT each) stay in flight per wave, ~22 KiB whatever
  // NT is: a CU needs >= ~48 KiB in flight to keep its share of the HBM stream (latency ~2 us under load)
  constexpr int D = 12 / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* a_img = reinterpret_cast<const uint4*>(smem);   // [k-step][row tile][row half][64 slots]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lc = lane & 15, lr = lane >> 4;
  const int total_steps = k >> 7;
  const int kt0 = blockIdx.y * steps_per_split;
  const int nk = min(steps_per_split, total_steps - kt0);
  const int n0 = (blockIdx.x * 4 + wave) * (16 * NT);
  // lane (lr, lc) of column tile t owns column n0 + NT * lc + t: its NT outputs of a row are adjacent
  const uint8_t* b_src[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int col = n0 + NT * lc + t;
    col = col < n ? col : n - 1;
    b_src[t] = b + (int64_t)col * ldb + (int64_t)kt0 * 128 + 16 * lr;
  }
  uint4 bq[D][NT][2];
  auto load_b = [&](int kk, uint4 (&dst)[NT][2]) {
    const int kc = kk < nk ? kk : nk - 1;     // (clamped, not branched: beyond the range the last step is re-read)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (NTL) {
        dst[t][0] = load_nt16(b_src[t] + kc * 128);
        dst[t][1] = load_nt16(b_src[t] + kc * 128 + 64);
      } else {
        // default cache policy: the two halves of a column's 128-byte line are fetched by two instructions, and a
        // non-temporal first half does not leave the line in L2 for the second (FETCH_SIZE: 2.1x the weight bytes
        // with nt, profiles/r03_pmc_summary.txt)
        dst[t][0] = *reinterpret_cast<const uint4*>(b_src[t] + kc * 128);
        dst[t][1] = *reinterpret_cast<const uint4*>(b_src[t] + kc * 128 + 64);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) load_b(s, bq[s]);
  // activation image of this K range: copy q = (kk * MT + i) * 2 + j = rows 8 j .. 8 j + 7 of row tile i, the 128 bytes
  // of k-step kk — whole cache lines (lane -> row l >> 3, chunk (l & 7) ^ (row & 6): the rotation that keeps the
  // fragment reads below off each other's banks, as in gemm8_packed_kernel's AROW form; the 16 rows x 64 bytes copies
  // of the first version sent 64 separate 16-byte requests per instruction); dealt round-robin to the 4 waves
  {
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int copies = nk * 2 * MT;
    const int rm_row = lane >> 3, rm_chunk = (lane & 7) ^ ((lane >> 3) & 6);
    for (int q = wave; q < copies; q += 4) {
      const int j = q & 1, i = (q >> 1) % MT, kk = (q >> 1) / MT;
      int row = 16 * i + 8 * j + rm_row;
      row = row < m ? row : m - 1;           // rows >= m only feed accumulator rows that are never stored
      lds_dma16(a + (int64_t)row * lda + (int64_t)(kt0 + kk) * 128 + 16 * rm_chunk, lds_base + q * 1024);
    }
    lds_dma_wait<0>();
  }
  __syncthreads();

  typename Op::acc_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = typename Op::acc_t{0, 0, 0, 0};
  }
  // lane (lr, lc) of row tile i: row lc's bytes 16 lr .. (k slice 0) and 64 + 16 lr .. (slice 1) of the k-step
  const int a_frag = (lc >> 3) * 64 + 8 * (lc & 7) + (lr ^ (lc & 6));
  auto mma = [&](int kk, uint4 (&cur)[NT][2]) {
    const uint4* ap = a_img + (kk * MT) * 128;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const uint4 a0 = ap[i * 128 + a_frag], a1 = ap[i * 128 + (a_frag ^ 4)];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[i][t] = Op::run32(a0, a1, cur[t][0], cur[t][1], acc[i][t]);
    }
  };
  int kk = 0;
  for (; kk + D <= nk; kk += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      load_b(kk + s + D - 1, bq[(s + D - 1) % D]);
      __builtin_amdgcn_sched_barrier(0);
      mma(kk + s, bq[s]);
    }
  }
  // the last nk - kk < D steps are already in flight (or in registers): no further loads
#pragma unroll
  for (int s = 0; s < D - 1; ++s) {
    if (kk + s < nk) mma(kk + s, bq[s]);
  }

  // epilogue: NT adjacent columns per lane and row
  const int col0 = n0 + NT * lc;
  if (col0 >= n) return;
  const bool split = gridDim.y > 1;
  float bs[NT], bi[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = col0 + t < n ? col0 + t : n - 1;
    bs[t] = b_scales[b_per_col ? col : 0];
    bi[t] = bias ? to_f32(bias[col]) : 0.f;
  }
  const bool vec = (n % NT) == 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = i * 16 + 4 * lr + j;
      if (row >= m) continue;
      if (split) {
        typename Op::elem_t* dst = ws + ((int64_t)blockIdx.y * m + row) * n + col0;
        if (vec && NT == 4) {
          typename Op::acc_t v = {acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]};
          *reinterpret_cast<typename Op::acc_t*>(dst) = v;
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (col0 + t < n) dst[t] = acc[i][t][j];
        }
      } else {
        const float as = a_scales[a_per_row ? row : 0];
        T o[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) o[t] = out_cast<T>(Op::finish(acc[i][t][j], as, bs[t], bi[t]));
        T* dst = out + (int64_t)row * ldc + col0;
        if (vec && NT == 4 && (ldc & 3) == 0) {
          *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o);
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (col0 + t < n) dst[t] = o[t];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- prefill (M > 320)
// Same structure as the w4a16 prefill GEMM (w4a16_unfused.hip).  int8, and fp8 with K % 128 != 0: both operands are
// first re-tiled into 1-KiB operand images (pack_a_kernel on the byte matrices viewed as 2-byte elements: a
// piece = 16 rows x 64 k-bytes, slot frag_swz(lr, lc) = row lc, bytes 16 lr .. 16 lr + 15), then a
// 256 x 256 tile / 8-wave kernel streams pieces by LDS-DMA through a 4-stage ring; a 16-byte
// fragment feeds two MFMAs (low / high 8 bytes — the same k permutation on both operands).
// P = 64-k pieces per stage and operand tile: 1 -> a 4-stage ring of 32-KiB stages, every 16-byte
// fragment feeds Op::run16; 2 (fp8, K % 128 == 0) -> two 64-KiB stages, the fragments of two
// adjacent pieces form the 32-byte operands of one 16x16x128 MFMA (Op::run32).
// IL: B was packed with interleaved rows (n % 64 == 0): a lane owns 4 adjacent output columns and
// stores them as one 8-byte word; otherwise lane lc = one column per 16-column tile.
// AROW / BROW (P = 2 only): the operand is NOT an image but the row-major byte matrix itself (leading dimension
// lda / ldb, 16-byte aligned rows) — no pack launch, no load-time weight image.  The LDS-DMA copies build the LDS image
// on the fly with per-lane source addresses: a wave-instruction = 8 rows x 128 bytes (whole cache lines: both 64-k
// slices of the stage; 16 rows x 64 bytes, half a line per request, cost the chunked-prefill job 4 % whenever the
// activations were not L2-hot, profiles/r03_fp8_operands_in_place.txt), slot 8 (row & 7) + (chunk ^ (row & 6)) with
// chunk = 4 slice + lr — the rotation keeps the ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...) on 16 distinct
// 16-byte bank groups; rows 0-7 / 8-15 of a tile sit where the image form keeps its two slices.
template <typename T, typename Op, int P, bool IL, bool AROW = false, bool BROW = false>
__global__ __launch_bounds__(512, 2) void gemm8_packed_kernel(
    T* __restrict__ out, const uint4* __restrict__ pa, const uint4* __restrict__ pb,
    const float* __restrict__ a_scales, int a_per_row, const float* __restrict__ b_scales,
    int b_per_col, const T* __restrict__ bias, int m, int n, int k, int64_t ldc, int num_m_blocks,
    int num_tiles, int64_t lda, int64_t ldb, typename Op::elem_t* __restrict__ slabs, int sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  constexpr int S = P == 1 ? 4 : 2;       // ring depth
  constexpr int kPiece = 2 * 16 * 64;     // uint4 per 64-k slice: 16 A pieces + 16 B pieces = 32 KiB
  constexpr int kStage = P * kPiece;
  constexpr int kBOff = 16 * 64;
  // sk > 1 (few tiles: chunked-prefill sized M, narrow N): work item (tile, split) runs one K range and leaves its
  // 4-byte partial tile in slabs[split][m][n]; the finish kernel of the decode path adds the slabs in split order
  // and applies scales / bias (deterministic, no atomics).  Work items = num_tiles * sk, split-major.
  // P = 2: PERSISTENT workgroups — the grid is min(work items, one workgroup per CU), workgroup w runs items w,
  // w + gridDim.x, ... (gridDim.x % 8 == 0 there, so an item's XCD is the one the one-item-per-workgroup launch gives
  // it) and issues the FIRST stage of its next item in the last k iteration of the current one: the copies travel
  // while the epilogue stores, instead of a workgroup launch and a cold first stage per tile.  Worth 1.5 % of the GEMM
  // time (profiles/r03_fp8_prefill_persistent.txt): of the ~12 us a tile costs beyond its 1.5 us per 128-k stage most
  // is the epilogue itself, which the only resident workgroup of a CU cannot overlap with MFMAs.
  const int total_work = num_tiles * sk;
  auto decode = [&](int work, int& split, int& mb, int& nb) {
    split = work / num_tiles;
    int tile;
    {
      const int b = work - split * num_tiles;
      const int q = num_tiles / 8, r = num_tiles % 8;
      const int xcd = b % 8;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    constexpr int GM = 8;
    const int num_n_blocks = num_tiles / num_m_blocks;
    const int group = tile / (GM * num_n_blocks);
    const int first_m = group * GM;
    const int gsz = min(num_m_blocks - first_m, GM);
    const int within = tile - group * GM * num_n_blocks;
    mb = first_m + within % gsz;
    nb = within / gsz;
  };
  int split, mb, nb;
  decode(blockIdx.x, split, mb, nb);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lc = lane & 15, lr = lane >> 4;
  const int kpieces = k / 64;       // pieces along K
  const int kstages_all = kpieces / P;
  const int per_split = (kstages_all + sk - 1) / sk;
  int ks0 = split * per_split;                                    // first stage of this work item's K range
  int kstages = max(0, min(per_split, kstages_all - ks0));

  static_assert(P == 2 || !(AROW || BROW), "operands in place: the two-slice stage only");
  // [i][j]: row tile i of this wave; j = k slice of the stage (image) or row half 0-7 / 8-15 (row-major)
  const uint4* a_src[2][P];
  const uint4* b_src[2][P];
  // row-major operands: lane -> (row of the half tile, 16-byte chunk of the stage's 128 k bytes)
  const int rm_row = lane >> 3, rm_chunk = (lane & 7) ^ ((lane >> 3) & 6);
  auto setup = [&](int mb, int nb, const uint4* (&as)[2][P], const uint4* (&bs)[2][P]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 2 + i;
    if constexpr (AROW) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        int row = mb * 256 + p * 16 + 8 * j + rm_row;
        row = row < m ? row : m - 1;          // rows >= m only feed accumulator rows that are never stored
        as[i][j] =
            reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(pa) + (int64_t)row * lda) + rm_chunk;
      }
    } else {
      int gmt = mb * 16 + p;
      const int max_mt = ((m + 15) >> 4) - 1;
      gmt = gmt < max_mt ? gmt : max_mt;
#pragma unroll
      for (int j = 0; j < P; ++j) as[i][j] = pa + ((int64_t)gmt * kpieces + j) * 64 + lane;
    }
    if constexpr (BROW) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        // IL: piece 4 G + t holds columns 64 G + 4 r + t (what pack_a_kernel's interleave does)
        const int r = 8 * j + rm_row;
        int col = nb * 256 + (IL ? 64 * (p >> 2) + 4 * r + (p & 3) : p * 16 + r);
        col = col < n ? col : n - 1;
        bs[i][j] =
            reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(pb) + (int64_t)col * ldb) + rm_chunk;
      }
    } else {
      int gnt = nb * 16 + p;
      const int max_nt = ((n + 15) >> 4) - 1;
      gnt = gnt < max_nt ? gnt : max_nt;
#pragma unroll
      for (int j = 0; j < P; ++j) bs[i][j] = pb + ((int64_t)gnt * kpieces + j) * 64 + lane;
    }
  }
  };
  setup(mb, nb, a_src, b_src);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto stage_from = [&](int buf, int ks_abs, const uint4* (&as)[2][P], const uint4* (&bs)[2][P]) {
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = wave * 2 + i;
        // a stage advances an image by P pieces of 64 slots, a row by P x 64 bytes = 4 P uint4
        const int dst = buf * kStage + pp * kPiece + p * 64;
        lds_dma16(as[i][pp] + (int64_t)ks_abs * (AROW ? 4 * P : 64 * P), lds_base + dst * 16);
        lds_dma16(bs[i][pp] + (int64_t)ks_abs * (BROW ? 4 * P : 64 * P), lds_base + (dst + kBOff) * 16);
      }
    }
  };
  auto stage = [&](int buf, int ks) { stage_from(buf, ks0 + ks, a_src, b_src); };
  typename Op::acc_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = typename Op::acc_t{0, 0, 0, 0};
  }
  if constexpr (P == 1) {
#pragma unroll
    for (int s2 = 0; s2 < S - 1; ++s2) {
      if (s2 < kstages) stage(s2, s2);
    }
  }
  const int frag = frag_swz(lr, lc);
  // P = 2: where a lane finds its 16 bytes of k slice 0 / 1 inside the tile's two KiB
  const int rm2 = (lc >> 3) * kPiece + 8 * (lc & 7) + (lr ^ (lc & 6));
  const int frag_a0 = AROW ? rm2 : frag, frag_a1 = AROW ? (rm2 ^ 4) : kPiece + frag;
  const int frag_b0 = BROW ? rm2 : frag, frag_b1 = BROW ? (rm2 ^ 4) : kPiece + frag;
  int cur = 0;
  auto epilogue = [&]() {
  if (sk > 1) {
    // partial tile -> slab of this split (raw accumulators; rows / columns as in the final epilogues below)
    typename Op::elem_t* sl = slabs + (int64_t)split * m * n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * 256 + wm * 128 + i * 16 + 4 * lr + j;
        if (row >= m) continue;
        if constexpr (IL) {
          const int col = nb * 256 + wn * 64 + 4 * lc;
          if (col < n) {
            const typename Op::acc_t v = {acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]};
            *reinterpret_cast<typename Op::acc_t*>(sl + (int64_t)row * n + col) = v;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int col = nb * 256 + (wn * 4 + t) * 16 + lc;
            if (col < n) sl[(int64_t)row * n + col] = acc[i][t][j];
          }
        }
      }
    }
    return;
  }
  if constexpr (IL) {
    const int col = nb * 256 + wn * 64 + 4 * lc;
    if (col >= n) return;
    float bs[4], bi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bs[t] = b_scales[b_per_col ? col + t : 0];
      bi[t] = bias ? to_f32(bias[col + t]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * 256 + wm * 128 + i * 16 + 4 * lr + j;
        if (row < m) {
          const float as = a_scales[a_per_row ? row : 0];
          T o[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = out_cast<T>(Op::finish(acc[i][t][j], as, bs[t], bi[t]));
          *reinterpret_cast<uint2*>(out + (int64_t)row * ldc + col) =
              *reinterpret_cast<const uint2*>(o);
        }
      }
    }
    return;
  }
  // epilogue: tile t of a wave = 16 consecutive columns (plain packing of B), lane lc = column
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = nb * 256 + (wn * 4 + t) * 16 + lc;
    if (col >= n) continue;
    const float bs = b_scales[b_per_col ? col : 0];
    const float bi = bias ? to_f32(bias[col]) : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * 256 + wm * 128 + i * 16 + 4 * lr + j;
        if (row < m) {
          const float as = a_scales[a_per_row ? row : 0];
          out[(int64_t)row * ldc + col] = out_cast<T>(Op::finish(acc[i][t][j], as, bs, bi));
        }
      }
    }
  }
  };
  if constexpr (P == 1) {
    // ping-pong of the two waves of a SIMD, as in gemm_packed_kernel (w4a16_unfused.hip): waves 4-7 enter
    // the loop one barrier late; an iteration is {12 fragment reads of stage ks, copies of stage ks+S-1,
    // own copies of stage ks+1 retired} barrier {32 MFMAs} barrier.
    if (kstages >= S - 1) lds_dma_wait<4 * (S - 2)>();
    else lds_dma_wait<0>();
    __syncthreads();
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    for (int ks = 0; ks < kstages; ++ks) {
      const uint4* abuf = lds + cur * kStage + (wm * 8) * 64 + frag;
      const uint4* bbuf = lds + cur * kStage + kBOff + (wn * 4) * 64 + frag;
      uint4 bf[4], af[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[t] = bbuf[t * 64];
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = abuf[i * 64];
      __builtin_amdgcn_sched_barrier(0);
      const int nxt = ks + S - 1;
      int slot = cur + S - 1;
      slot = slot >= S ? slot - S : slot;
      if (nxt < kstages) {
        stage(slot, nxt);
        lds_dma_wait<4 * (S - 2)>();
      } else {
        lds_dma_wait<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = Op::run16(af[i], bf[t], acc[i][t]);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      cur = cur + 1 == S ? 0 : cur + 1;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    epilogue();
  } else {
    // two 64-KiB stages, all 8 waves in phase, one barrier per stage.  (The ping-pong needs a copy's
    // issue -> retire window to span a barrier epoch pair; with only two slots both groups' copies of stage
    // ks+1 would have to go out and retire within epochs 2ks .. 2ks+1 — built and measured 4 % SLOWER than
    // this loop, 1803 vs 1727 us per Llama-3-8B layer at M = 8192.)
    int work = blockIdx.x;
    bool prefetched = false;      // the first stage of the current item is already on its way into slot `cur`
    bool first = true;            // (a_src / b_src were set up for it above)
    for (;;) {
      // the next item of this workgroup, if any (scalars only: its pointers replace a_src / b_src in the last k
      // iteration, when the current item has no stage left to issue)
      const int nwork = work + (int)gridDim.x;
      const bool has_next = nwork < total_work;
      int nsplit = 0, nmb = 0, nnb = 0, nks0 = 0, nkstages = 0;
      if (has_next) {
        decode(nwork, nsplit, nmb, nnb);
        nks0 = nsplit * per_split;
        nkstages = max(0, min(per_split, kstages_all - nks0));
      }
      if (!prefetched) {
        // (slot `cur` is free: its last readers passed the barrier of the previous item's last iteration)
        if (!first) setup(mb, nb, a_src, b_src);
        if (kstages > 0) stage(cur, 0);
      }
      prefetched = false;
      first = false;
      for (int ks = 0; ks < kstages; ++ks) {
        lds_dma_wait<0>();
        __syncthreads();
        {
          const int slot = cur ^ 1;
          if (ks + 1 < kstages) {
            stage(slot, ks + 1);
          } else if (nkstages > 0) {
            setup(nmb, nnb, a_src, b_src);
            stage_from(slot, nks0, a_src, b_src);
            prefetched = true;
          }
        }
        const uint4* abuf = lds + cur * kStage + (wm * 8) * 64;
        const uint4* bbuf = lds + cur * kStage + kBOff + (wn * 4) * 64;
        uint4 bf[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bf[t][0] = bbuf[t * 64 + frag_b0];
          bf[t][1] = bbuf[t * 64 + frag_b1];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4 af[4][2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            af[i][0] = abuf[(4 * h + i) * 64 + frag_a0];
            af[i][1] = abuf[(4 * h + i) * 64 + frag_a1];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              acc[4 * h + i][t] =
                  Op::run32(af[i][0], af[i][1], bf[t][0], bf[t][1], acc[4 * h + i][t]);
            }
          }
        }
        cur ^= 1;
      }
      epilogue();
      if (!has_next) break;
      work = nwork;
      split = nsplit; mb = nmb; nb = nnb; ks0 = nks0; kstages = nkstages;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = typename Op::acc_t{0, 0, 0, 0};
      }
    }
  }
}

// split-K epilogue: out = T((slab 0 + slab 1 + ...) * a_s * b_s + bias), slabs added in index order
template <typename T, typename Op>
__global__ void fp8_gemm_finish_kernel(T* __restrict__ out, const typename Op::elem_t* __restrict__ ws,
                                       const float* __restrict__ a_scales, int a_per_row,
                                       const float* __restrict__ b_scales, int b_per_col,
                                       const T* __restrict__ bias, int m, int n, int64_t ldc, int sk) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (col >= n) return;
  typename Op::elem_t acc = ws[(int64_t)row * n + col];
  for (int s = 1; s < sk; ++s) acc += ws[((int64_t)s * m + row) * n + col];
  const float v = Op::finish(acc, a_scales[a_per_row ? row : 0],
                             b_scales[b_per_col ? col : 0], bias ? to_f32(bias[col]) : 0.f);
  out[(int64_t)row * ldc + col] = out_cast<T>(v);
}
// the same for n % 4 == 0, 8-byte-aligned rows of `out`: one thread = 4 adjacent columns of one row (16-byte slab
// loads, one 8-byte store), the same additions in the same order
template <typename T, typename Op>
__global__ __launch_bounds__(256) void fp8_gemm_finish4_kernel(
    T* __restrict__ out, const typename Op::elem_t* __restrict__ ws, const float* __restrict__ a_scales,
    int a_per_row, const float* __restrict__ b_scales, int b_per_col, const T* __restrict__ bias, int m, int n,
    int64_t ldc, int sk) {
  const int n4 = n >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= m * n4) return;
  const int row = idx / n4, col = (idx - row * n4) * 4;
  typedef typename Op::acc_t vec_t;
  vec_t acc = *reinterpret_cast<const vec_t*>(ws + (int64_t)row * n + col);
  for (int s = 1; s < sk; ++s) {
    const vec_t v = *reinterpret_cast<const vec_t*>(ws + ((int64_t)s * m + row) * n + col);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  const float as = a_scales[a_per_row ? row : 0];
  T o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    o[t] = out_cast<T>(Op::finish(acc[t], as, b_scales[b_per_col ? col + t : 0], bias ? to_f32(bias[col + t]) : 0.f));
  *reinterpret_cast<uint2*>(out + (int64_t)row * ldc + col) = *reinterpret_cast<const uint2*>(o);
}

// plan of the streaming decode kernel: NT column tiles per wave (a workgroup covers 64 NT columns), `sk` K
// splits of `steps` 128-byte k-steps.  Aims: >= ~1 workgroup per CU, a K range of >= 4 k-steps per workgroup
// where the shape allows it, as few partial slabs as that permits, and the range's activation image in LDS.
struct DecodePlan {
  int nt, sk, steps;
};
static bool plan_decode(int m, int n, int k, int64_t ws_elems, DecodePlan* p) {
  const int total = k / 128;
  const int mt = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
  const int max_steps = (128 * 1024) / (mt * 2048);     // activation image <= 128 KiB
  if (const char* f = getenv("MI355X_F8_DECODE_FORCE")) {   // "nt,sk" (kernel experiments)
    int nt = 4, sk = 1;
    if (sscanf(f, "%d,%d", &nt, &sk) == 2 && (nt == 1 || nt == 2 || nt == 4) && sk >= 1) {
      if (sk > total) sk = total;
      int steps = (total + sk - 1) / sk;
      sk = (total + steps - 1) / steps;
      if (steps <= max_steps && (sk == 1 || (int64_t)sk * m * n <= ws_elems)) {
        *p = DecodePlan{nt, sk, steps};
        return true;
      }
    }
  }
  // cost model fitted to scripts/bench_scaled_mm_decode.py sweeps (Llama-3-70B per-rank and Llama-3-8B shapes at
  // M = 64): launch + first-round-trip latency, the activation image (L2 -> LDS at ~70 GB/s per CU), the weight
  // stream at a CU's share of ~6 TB/s (at most ~45 GB/s when few CUs work), and for a K split the finish launch
  // plus the slab round trip.  It reproduces the measured winners: one workgroup per CU, the smallest K split
  // that reaches ~200 workgroups, no split at all when N alone gives >= 128 of them.
  double best = 1e30;
  bool found = false;
  for (int nt = 4; nt >= 1; nt >>= 1) {
    const int nblk = (n + 64 * nt - 1) / (64 * nt);
    for (int sk = 1; sk <= 16 && sk <= total; ++sk) {
      const int steps = (total + sk - 1) / sk;
      if ((total + steps - 1) / steps != sk || steps > max_steps) continue;
      if (sk > 1 && (int64_t)sk * m * n > ws_elems) continue;
      const int wgs = nblk * sk;
      const int img = mt * steps * 2048;
      int per_cu = (160 * 1024) / img;
      per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
      const int rounds = (wgs + 256 * per_cu - 1) / (256 * per_cu);
      const int active = wgs < 256 ? wgs : 256;
      double rate = 6000.0 / active;              // GB/s per CU
      if (rate > 45.0) rate = 45.0;
      int conc = (wgs + 255) / 256;
      if (conc > per_cu) conc = per_cu;
      const double step_us = 8.2 * nt / rate * conc;
      double t = 5.5 + rounds * (img / 1024.0 / 68.0 + steps * step_us);
      if (sk > 1) t += 3.5 + 2.0 * (double)sk * m * n * 4 / 4e6;
      if (t < best) {
        best = t;
        *p = DecodePlan{nt, sk, steps};
        found = true;
      }
    }
  }
  return found;
}

template <typename T, typename Op, int MT, int NT>
static int launch_decode(const Fp8Args& g, int a_per_row, int b_per_col, const DecodePlan& p) {
  static const bool nt_loads = [] { const char* e = getenv("MI355X_F8_NT"); return e && e[0] == '1'; }();
  auto kern = nt_loads ? gemm8_decode_kernel<T, Op, MT, NT, true> : gemm8_decode_kernel<T, Op, MT, NT, false>;
  const size_t smem = (size_t)MT * p.steps * 2048;
  static PerDeviceOnce once;   // one per instantiation
  int dev;
  if (once.need(&dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_decode_kernel<T, Op, MT, NT, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_decode_kernel<T, Op, MT, NT, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) {
      set_error("scaled_mm(decode): cannot reserve 128 KiB of LDS: %s", hipGetErrorString(e));
      return MI355X_EUNSUPPORTED;
    }
    once.mark(dev);
  }
  const int nblk = (g.n + 64 * NT - 1) / (64 * NT);
  hipLaunchKernelGGL(kern, dim3(nblk, p.sk), dim3(256), smem, g.stream, static_cast<T*>(g.out),
                     reinterpret_cast<typename Op::elem_t*>(g.ws), g.a, g.b, g.a_scales, a_per_row, g.b_scales,
                     b_per_col, static_cast<const T*>(g.bias), g.m, g.n, g.k, g.lda, g.ldb, g.ldc, p.steps);
  int rc = check_launch("scaled_mm(decode)");
  if (rc || p.sk == 1) return rc;
  if (g.defer_sk != nullptr) {   // the consumer adds the slabs and applies the scales (common.cuh: slab_values)
    *g.defer_sk = p.sk;
    return MI355X_OK;
  }
  if (g.n % 4 == 0 && g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.out) & 7) == 0 &&
      (reinterpret_cast<uintptr_t>(g.ws) & 15) == 0) {
    hipLaunchKernelGGL((fp8_gemm_finish4_kernel<T, Op>), dim3((g.m * (g.n / 4) + 255) / 256), dim3(256), 0, g.stream,
                       static_cast<T*>(g.out), reinterpret_cast<const typename Op::elem_t*>(g.ws), g.a_scales,
                       a_per_row, g.b_scales, b_per_col, static_cast<const T*>(g.bias), g.m, g.n, g.ldc, p.sk);
  } else {
    hipLaunchKernelGGL((fp8_gemm_finish_kernel<T, Op>), dim3((g.n + 255) / 256, g.m), dim3(256), 0, g.stream,
                       static_cast<T*>(g.out), reinterpret_cast<const typename Op::elem_t*>(g.ws), g.a_scales,
                       a_per_row, g.b_scales, b_per_col, static_cast<const T*>(g.bias), g.m, g.n, g.ldc, p.sk);
  }
  return check_launch("scaled_mm(decode finish)");
}

// 0 / negative: done / error; 1: not a shape this path takes
template <typename T, typename Op>
static int run_decode(const Fp8Args& g, int a_per_row, int b_per_col) {
  DecodePlan p;
  if (!plan_decode(g.m, g.n, g.k, g.ws ? g.ws_elems : 0, &p)) return 1;
#define DEC_NT(MTV)                                                                   \
  (p.nt == 4 ? launch_decode<T, Op, MTV, 4>(g, a_per_row, b_per_col, p)               \
             : (p.nt == 2 ? launch_decode<T, Op, MTV, 2>(g, a_per_row, b_per_col, p)  \
                          : launch_decode<T, Op, MTV, 1>(g, a_per_row, b_per_col, p)))
  if (g.m <= 16) return DEC_NT(1);
  if (g.m <= 32) return DEC_NT(2);
  return DEC_NT(4);
#undef DEC_NT
}

// K split of the prefill kernel (gemm8_packed_kernel): plan_tile_split (w4a16.cuh); MI355X_F8_PACKED_SK for A/B runs
static int plan_packed_split(int m, int n, int k, int64_t max_slab_elems) {
  static const int forced = [] { const char* e = getenv("MI355X_F8_PACKED_SK"); return e ? atoi(e) : 0; }();
  return plan_tile_split(m, n, k / 128, max_slab_elems, forced);
}

static int f8_rowmajor_bits() {
  static const int bits = [] { const char* e = getenv("MI355X_F8_ROWMAJOR"); return e ? atoi(e) : 3; }();
  return bits;
}

template <typename T, typename Op>
static int run_fp8(const Fp8Args& g) {
  const int a_per_row = g.a_scales_numel > 1;
  const int b_per_col = g.b_scales_numel > 1;
  T* out = static_cast<T*>(g.out);
  const T* bias = static_cast<const T*>(g.bias);
  {
    // prefill: both operands re-tiled into operand images, then the LDS-DMA ring kernel
    const int64_t m_pad = ((int64_t)g.m + 15) / 16 * 16, n_pad = ((int64_t)g.n + 15) / 16 * 16;
    // fp8, K % 128 == 0 (the two-slice stage): operands with 16-byte aligned rows are read in place — no pack launch,
    // and weights without a load-time image cost nothing extra (Llama-3-8B layer at M = 576 / 1024 / 4096 / 8192:
    // 396 / 409 / 855 / 1541 us with both images -> 374 / 383 / 798 / 1452 with the activations in place, 374 / 380 /
    // 808 / 1466 with both in place).  MI355X_F8_ROWMAJOR (bit 0: activations, bit 1: weights) for A/B runs.
    const int kRowMajor = f8_rowmajor_bits();
    bool wide = false;
    if constexpr (Op::kWide) wide = g.k % 128 == 0;
    const bool arow = wide && (kRowMajor & 1) && g.lda % 16 == 0 && (reinterpret_cast<uintptr_t>(g.a) & 15) == 0;
    const bool brow = arow && (kRowMajor & 2) && !g.b_image && g.ldb % 16 == 0 &&
                      (reinterpret_cast<uintptr_t>(g.b) & 15) == 0;
    const int64_t need = (arow ? 0 : m_pad * g.k) + (g.b_image || brow ? 0 : n_pad * g.k);   // bytes
    // 320 < M < 1024 too since round 3 (was the direct-load tile kernel: Llama-3-8B layer at M = 384 / 512 / 768
    // 660 / 672 / 772 -> 409 / 422 / 470 us, profiles/r03_scaled_mm_mid_m.txt); MI355X_F8_PACKED_MIN_M for A/B runs
    static const int kPackedMinM = [] { const char* e = getenv("MI355X_F8_PACKED_MIN_M"); return e ? atoi(e) : 321; }();
    if (g.m >= kPackedMinM && (need == 0 || (g.ws != nullptr && g.ws_elems * 4 >= need &&
                                             (reinterpret_cast<uintptr_t>(g.ws) & 15) == 0)) &&
        g.k % 64 == 0) {
      // what the workspace holds beyond the operand images: slabs of a K split
      typename Op::elem_t* slabs = nullptr;
      int sk = 1;
      if (g.ws != nullptr && (reinterpret_cast<uintptr_t>(g.ws) & 15) == 0) {
        const int64_t img_elems = (need + 15) / 16 * 4;
        sk = plan_packed_split(g.m, g.n, g.k, g.ws_elems - img_elems);
        slabs = reinterpret_cast<typename Op::elem_t*>(g.ws) + img_elems;
      }
      bf16_t* ws16 = reinterpret_cast<bf16_t*>(g.ws);
      const void* pa = arow ? g.a : static_cast<const void*>(ws16);
      bf16_t* pb_ws = arow ? ws16 : ws16 + m_pad * g.k / 2;
      const void* pb = g.b_image ? g.b_image : (brow ? g.b : static_cast<const void*>(pb_ws));
      // 4 adjacent output columns per lane (8-byte stores) when the shape allows it (a prepacked image is always
      // interleaved: its entry point checked the same conditions)
      const bool il = g.n % 64 == 0 && g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;
      const int k2 = g.k / 2;   // the byte matrices viewed as 2-byte elements
      if (!arow)
        hipLaunchKernelGGL(pack_a_kernel<bf16_t>, dim3((k2 + kPackK - 1) / kPackK, (int)(m_pad / 16)),
                           dim3(256), 0, g.stream, ws16, reinterpret_cast<const bf16_t*>(g.a), g.m, k2,
                           g.lda / 2);
      if (!g.b_image && !brow)
        hipLaunchKernelGGL(pack_a_kernel<bf16_t>, dim3((k2 + kPackK - 1) / kPackK, (int)(n_pad / 16)),
                           dim3(256), 0, g.stream, pb_ws, reinterpret_cast<const bf16_t*>(g.b), g.n, k2,
                           g.ldb / 2, il ? 1 : 0);
      int rc = check_launch("scaled_mm(pack)");
      if (rc) return rc;
      const int num_m_blocks = (g.m + 255) / 256, num_n_blocks = (g.n + 255) / 256;
      const int num_tiles = num_m_blocks * num_n_blocks;
      const size_t smem = (size_t)4 * 2048 * sizeof(uint4);   // 128 KiB in either ring shape
      // the two-slice kernel runs persistent workgroups, one per CU (128 KiB of LDS each: a CU holds one), over the
      // (tile, split) items; MI355X_F8_PERSIST=0: one workgroup per item, for A/B runs
      static const int persistent_wgs = [] {
        const char* e = getenv("MI355X_F8_PERSIST");
        if (e && e[0] == '0') return 1 << 30;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
          cus = 256;
        return cus >= 8 ? cus / 8 * 8 : 8;     // (a multiple of 8: an item keeps the XCD of its index)
      }();
      auto launch = [&](auto kern, PerDeviceOnce& once) -> int {
        int dev;
        if (once.need(&dev)) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
          if (e != hipSuccess) {
            set_error("scaled_mm(packed): cannot reserve %zu B of LDS: %s", smem,
                      hipGetErrorString(e));
            return MI355X_EUNSUPPORTED;
          }
          once.mark(dev);
        }
        // (persistent from two rounds of items on: below that there is nothing to chain, and the chunked-prefill job
        //  ran 3 % slower with 256 workgroups over 336 items, profiles/r03_fp8_prefill_persistent.txt)
        const int items = num_tiles * sk;
        hipLaunchKernelGGL(kern, dim3(wide && items >= 2 * persistent_wgs ? persistent_wgs : items), dim3(512), smem,
                           g.stream, out,
                           reinterpret_cast<const uint4*>(pa), reinterpret_cast<const uint4*>(pb),
                           g.a_scales, a_per_row, g.b_scales, b_per_col, bias, g.m, g.n, g.k, g.ldc,
                           num_m_blocks, num_tiles, g.lda, g.ldb, slabs, sk);
        return 0;
      };
      static PerDeviceOnce attr[8];   // per kernel variant, one bit per device
      if constexpr (Op::kWide) {
        if (wide) {
          if (brow) rc = il ? launch(gemm8_packed_kernel<T, Op, 2, true, true, true>, attr[7])
                            : launch(gemm8_packed_kernel<T, Op, 2, false, true, true>, attr[6]);
          else if (arow) rc = il ? launch(gemm8_packed_kernel<T, Op, 2, true, true, false>, attr[5])
                                 : launch(gemm8_packed_kernel<T, Op, 2, false, true, false>, attr[4]);
          else rc = il ? launch(gemm8_packed_kernel<T, Op, 2, true>, attr[3])
                       : launch(gemm8_packed_kernel<T, Op, 2, false>, attr[2]);
        }
      }
      if (!wide) {
        rc = il ? launch(gemm8_packed_kernel<T, Op, 1, true>, attr[1])
                : launch(gemm8_packed_kernel<T, Op, 1, false>, attr[0]);
      }
      if (rc) return rc;
      rc = check_launch("scaled_mm(packed)");
      if (rc || sk == 1) return rc;
      if (g.n % 4 == 0 && g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
        hipLaunchKernelGGL((fp8_gemm_finish4_kernel<T, Op>), dim3((g.m * (g.n / 4) + 255) / 256), dim3(256), 0,
                           g.stream, out, slabs, g.a_scales, a_per_row, g.b_scales, b_per_col, bias, g.m, g.n, g.ldc, sk);
      } else {
        hipLaunchKernelGGL((fp8_gemm_finish_kernel<T, Op>), dim3((g.n + 255) / 256, g.m), dim3(256), 0, g.stream, out,
                           slabs, g.a_scales, a_per_row, g.b_scales, b_per_col, bias, g.m, g.n, g.ldc, sk);
      }
      return check_launch("scaled_mm(packed finish)");
    }
  }
  // 64 < M <= 320 (decode batches above 64 sequences, small chunked-prefill budgets): passes of 64 rows through the
  // streaming decode kernel — the weights of the second and later passes come from L2 / MALL.  Llama-3-8B layer at
  // M = 128 / 256: 607 / 617 us on the direct-load tile kernel below (16-48 workgroups for the narrow projections,
  // profiles/r03_scaled_mm_mid_m.txt) against ~100 us per pass here.
  if (g.m > 64 && g.m <= 320 && g.k % 128 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && g.ldc % 4 == 0 &&
      !getenv("MI355X_F8_DECODE_OLD")) {
    bool done = true;
    for (int row0 = 0; row0 < g.m; row0 += 64) {
      Fp8Args p = g;
      p.m = g.m - row0 < 64 ? g.m - row0 : 64;
      p.a = g.a + (int64_t)row0 * g.lda;
      p.out = static_cast<T*>(g.out) + (int64_t)row0 * g.ldc;
      if (a_per_row) {
        p.a_scales = g.a_scales + row0;
        p.a_scales_numel = p.m > 1 ? p.m : 2;   // (stays "per row" for a one-row tail)
      }
      p.defer_sk = nullptr;
      const int rc = run_decode<T, Op>(p, a_per_row, b_per_col);
      if (rc < 0) return rc;
      if (rc == 1) {           // no plan for this shape: only possible before anything was launched
        done = false;
        break;
      }
    }
    if (done) return MI355X_OK;
  }
  if (g.m > 64) {
    const int num_m_blocks = (g.m + kF8BM - 1) / kF8BM;
    const int num_n_blocks = (g.n + kF8BN - 1) / kF8BN;
    const int num_tiles = num_m_blocks * num_n_blocks;
    hipLaunchKernelGGL((fp8_gemm_large_kernel<T, Op>), dim3(num_tiles), dim3(256),
                       (size_t)2 * 8 * 64 * sizeof(uint4), g.stream, out, g.a, g.b, g.a_scales,
                       a_per_row, g.b_scales, b_per_col, bias, g.m, g.n, g.k, g.lda, g.ldb, g.ldc,
                       num_m_blocks, num_tiles);
    return check_launch("scaled_mm_fp8(large)");
  }
  if (g.k % 128 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && !getenv("MI355X_F8_DECODE_OLD")) {
    const int rc = run_decode<T, Op>(g, a_per_row, b_per_col);
    if (rc <= 0) return rc;      // 1: no plan for this shape, the older kernel below takes it
  }
  const int col_tiles = (g.n + 63) / 64;
  const int total_tiles = g.k / kF8BK;
  int sk = 1;
  if (g.ws != nullptr) {
    // one fp32 / int32 partial slab [m, n] per K split; only as many splits as the workspace holds
    while (col_tiles * sk < 256 && total_tiles / (sk * 2) >= 16 &&
           g.ws_elems >= (int64_t)(sk * 2) * g.m * g.n)
      sk *= 2;
  }
  const int per_split = (total_tiles + sk - 1) / sk;
  sk = (total_tiles + per_split - 1) / per_split;
  const int mt = (g.m + 15) / 16;
  dim3 grid(col_tiles, sk), block(256);
#define LAUNCH_F8S(MTV)                                                                       \
  hipLaunchKernelGGL((fp8_gemm_small_kernel<T, MTV, Op>), grid, block,                        \
                     (size_t)2 * MTV * 16 * 64 * sizeof(float), g.stream, out,                 \
                     reinterpret_cast<typename Op::elem_t*>(g.ws), g.a, g.b,                   \
                     g.a_scales, a_per_row, g.b_scales, b_per_col, bias, g.m, g.n, g.k, g.lda, \
                     g.ldb, g.ldc, per_split)
  if (mt <= 1) LAUNCH_F8S(1);
  else if (mt <= 2) LAUNCH_F8S(2);
  else LAUNCH_F8S(4);
#undef LAUNCH_F8S
  int rc = check_launch("scaled_mm_fp8(small)");
  if (rc || sk == 1) return rc;
  hipLaunchKernelGGL((fp8_gemm_finish_kernel<T, Op>), dim3((g.n + 255) / 256, g.m), dim3(256), 0,
                     g.stream, out, reinterpret_cast<const typename Op::elem_t*>(g.ws), g.a_scales, a_per_row,
                     g.b_scales, b_per_col, bias, g.m, g.n, g.ldc, sk);
  return check_launch("scaled_mm_fp8(finish)");
}

}  // namespace mi355x

using namespace mi355x;

extern "C" int mi355x_scaled_mm_fp8(void* out, const void* a, const void* b,
                                    const float* a_scales, int a_scales_numel,
                                    const float* b_scales, int b_scales_numel,
                                    const void* bias, float* workspace,
                                    int64_t workspace_elems, int m, int n, int k, int64_t lda,
                                    int64_t ldb, int64_t ldc, int out_dtype,
                                    mi355x_stream stream) {
  MI355X_REQUIRE(m >= 0 && n > 0 && k > 0, MI355X_EINVAL, "scaled_mm_fp8: bad sizes");
  MI355X_REQUIRE(k % 64 == 0, MI355X_EUNSUPPORTED, "scaled_mm_fp8: k = %d must be a multiple of 64", k);
  MI355X_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, MI355X_EUNSUPPORTED,
                 "scaled_mm_fp8: lda / ldb must be multiples of 16 bytes");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == m) &&
                     (b_scales_numel == 1 || b_scales_numel == n),
                 MI355X_EINVAL, "scaled_mm_fp8: scales must be per-tensor or per-row / per-column");
  if (m == 0) return MI355X_OK;
  MI355X_REQUIRE(out && a && b && a_scales && b_scales, MI355X_EINVAL, "scaled_mm_fp8: null pointer");
  MI355X_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0,
                 MI355X_EUNSUPPORTED, "scaled_mm_fp8: a and b must be 16-byte aligned");
  Fp8Args g{out, static_cast<const uint8_t*>(a), static_cast<const uint8_t*>(b), a_scales,
            a_scales_numel, b_scales, b_scales_numel, bias, m, n, k, lda, ldb, ldc, workspace,
            workspace_elems, static_cast<hipStream_t>(stream)};
  return MI355X_DISPATCH_HALF(out_dtype, [&] { return run_fp8<scalar_t, OpFp8>(g); });
}

// 4-byte workspace elements the prefill kernel (m > 320) wants ON TOP of the operand-image scratch to split K for a
// shape with few 256 x 256 tiles (0: no split) — what a binding adds to the workspace it passes to mi355x_scaled_mm_*.
extern "C" int64_t mi355x_scaled_mm_split_elems(int m, int n, int k) {
  if (m <= 320 || n <= 0 || k <= 0 || k % 64 != 0) return 0;
  const int sk = plan_packed_split(m, n, k, INT64_MAX);
  return sk > 1 ? (int64_t)sk * m * n : 0;
}

// Load-time weight image of the 8-bit GEMM's packed path (fp8 and int8 alike: bytes): what pack_a_kernel derives
// from `b` on every call with m > 320, computed once — n * k bytes, rows interleaved by 4 (n % 64 == 0, k % 64 == 0).
// Returns 1 when the shape has no such image (the caller keeps calling mi355x_scaled_mm_*).
extern "C" int mi355x_scaled_mm_prepack(void* image, const void* b, int n, int k, int64_t ldb,
                                        mi355x_stream stream) {
  MI355X_REQUIRE(n > 0 && k > 0, MI355X_EINVAL, "scaled_mm_prepack: bad sizes");
  MI355X_REQUIRE(image && b, MI355X_EINVAL, "scaled_mm_prepack: null pointer");
  if (n % 64 != 0 || k % 64 != 0 || ldb % 16 != 0 ||
      ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(b)) & 15) != 0)
    return 1;
  const int k2 = k / 2;
  hipLaunchKernelGGL(pack_a_kernel<bf16_t>, dim3((k2 + kPackK - 1) / kPackK, n / 16), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<bf16_t*>(image),
                     static_cast<const bf16_t*>(b), n, k2, ldb / 2, 1);
  return check_launch("scaled_mm_prepack");
}

// cutlass_scaled_mm with the weights given as that image (m > 320, the packed path; workspace: >= roundup(m, 16) * k
// bytes for the activation image — may be NULL for fp8 with k % 128 == 0, whose activations are read in place).
// is_int8: the int8 arithmetic of mi355x_scaled_mm_int8.  Bit-identical to the call on `b` itself.
extern "C" int mi355x_scaled_mm_prepacked(void* out, const void* a, const void* b_image,
                                          const float* a_scales, int a_scales_numel,
                                          const float* b_scales, int b_scales_numel, const void* bias,
                                          float* workspace, int64_t workspace_elems, int m, int n, int k,
                                          int64_t lda, int64_t ldc, int out_dtype, int is_int8,
                                          mi355x_stream stream) {
  MI355X_REQUIRE(m > 320 && n > 0 && k > 0 && n % 64 == 0 && k % 64 == 0, MI355X_EINVAL,
                 "scaled_mm_prepacked: needs m > 320, n %% 64 == 0, k %% 64 == 0");
  MI355X_REQUIRE(lda % 16 == 0 && ldc % 4 == 0, MI355X_EUNSUPPORTED, "scaled_mm_prepacked: lda %% 16, ldc %% 4");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == m) && (b_scales_numel == 1 || b_scales_numel == n),
                 MI355X_EINVAL, "scaled_mm_prepacked: scales must be per-tensor or per-row / per-column");
  MI355X_REQUIRE(out && a && b_image && a_scales && b_scales, MI355X_EINVAL, "scaled_mm_prepacked: null pointer");
  MI355X_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b_image) |
                   reinterpret_cast<uintptr_t>(workspace)) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                 MI355X_EUNSUPPORTED, "scaled_mm_prepacked: alignment");
  const int64_t m_pad = ((int64_t)m + 15) / 16 * 16;
  const bool a_in_place = !is_int8 && k % 128 == 0 && (f8_rowmajor_bits() & 1);
  MI355X_REQUIRE(a_in_place || (workspace && workspace_elems * 4 >= m_pad * k), MI355X_EINVAL,
                 "scaled_mm_prepacked: workspace too small");
  Fp8Args g{out, static_cast<const uint8_t*>(a), nullptr, a_scales, a_scales_numel, b_scales, b_scales_numel, bias,
            m, n, k, lda, 0, ldc, workspace, workspace_elems, static_cast<hipStream_t>(stream)};
  g.b_image = b_image;
  if (is_int8) return MI355X_DISPATCH_HALF(out_dtype, [&] { return run_fp8<scalar_t, OpI8>(g); });
  return MI355X_DISPATCH_HALF(out_dtype, [&] { return run_fp8<scalar_t, OpFp8>(g); });
}

// Decode-sized (m <= 64) fp8 GEMM whose K split, if the plan has one, is left as fp32 partial slabs
// workspace[sk][m][n] for the consumer (*sk_out = sk; the consumer computes T(sum * a_s * b_s) = what this GEMM's
// finish kernel stores — mi355x_paged_attention_fused_qkv_w8, mi355x_silu_and_mul_per_token_quant_slabs,
// mi355x_rms_norm_dynamic_per_token_quant_slabs).  *sk_out = 0: `out` is final (no split, or a shape the streaming
// kernel does not take).  No bias.
extern "C" int mi355x_scaled_mm_fp8_deferred(void* out, const void* a, const void* b,
                                             const float* a_scales, int a_scales_numel,
                                             const float* b_scales, int b_scales_numel,
                                             float* workspace, int64_t workspace_elems, int m, int n, int k,
                                             int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int* sk_out,
                                             mi355x_stream stream) {
  MI355X_REQUIRE(sk_out != nullptr, MI355X_EINVAL, "scaled_mm_fp8_deferred: null sk_out");
  *sk_out = 0;
  MI355X_REQUIRE(m >= 0 && n > 0 && k > 0, MI355X_EINVAL, "scaled_mm_fp8_deferred: bad sizes");
  MI355X_REQUIRE(k % 64 == 0, MI355X_EUNSUPPORTED, "scaled_mm_fp8_deferred: k = %d must be a multiple of 64", k);
  MI355X_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, MI355X_EUNSUPPORTED,
                 "scaled_mm_fp8_deferred: lda / ldb must be multiples of 16 bytes");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == m) && (b_scales_numel == 1 || b_scales_numel == n),
                 MI355X_EINVAL, "scaled_mm_fp8_deferred: scales must be per-tensor or per-row / per-column");
  if (m == 0) return MI355X_OK;
  MI355X_REQUIRE(out && a && b && a_scales && b_scales, MI355X_EINVAL, "scaled_mm_fp8_deferred: null pointer");
  MI355X_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0,
                 MI355X_EUNSUPPORTED, "scaled_mm_fp8_deferred: a and b must be 16-byte aligned");
  Fp8Args g{out, static_cast<const uint8_t*>(a), static_cast<const uint8_t*>(b), a_scales,
            a_scales_numel, b_scales, b_scales_numel, nullptr, m, n, k, lda, ldb, ldc, workspace,
            workspace_elems, static_cast<hipStream_t>(stream)};
  // slabs are consumed as 16-byte pieces of whole rows: n % 8 == 0, aligned scratch
  if (m <= 64 && n % 8 == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) g.defer_sk = sk_out;
  return MI355X_DISPATCH_HALF(out_dtype, [&] { return run_fp8<scalar_t, OpFp8>(g); });
}

extern "C" int mi355x_scaled_mm_int8(void* out, const void* a, const void* b,
                                     const float* a_scales, int a_scales_numel,
                                     const float* b_scales, int b_scales_numel,
                                     const void* bias, float* workspace,
                                     int64_t workspace_elems, int m, int n, int k, int64_t lda,
                                     int64_t ldb, int64_t ldc, int out_dtype,
                                     mi355x_stream stream) {
  MI355X_REQUIRE(m >= 0 && n > 0 && k > 0, MI355X_EINVAL, "scaled_mm_int8: bad sizes");
  MI355X_REQUIRE(k % 64 == 0, MI355X_EUNSUPPORTED, "scaled_mm_int8: k = %d must be a multiple of 64", k);
  MI355X_REQUIRE(lda % 16 == 0 && ldb % 16 == 0, MI355X_EUNSUPPORTED,
                 "scaled_mm_int8: lda / ldb must be multiples of 16 bytes");
  MI355X_REQUIRE((a_scales_numel == 1 || a_scales_numel == m) &&
                     (b_scales_numel == 1 || b_scales_numel == n),
                 MI355X_EINVAL, "scaled_mm_int8: scales must be per-tensor or per-row / per-column");
  if (m == 0) return MI355X_OK;
  MI355X_REQUIRE(out && a && b && a_scales && b_scales, MI355X_EINVAL, "scaled_mm_int8: null pointer");
  MI355X_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0,
                 MI355X_EUNSUPPORTED, "scaled_mm_int8: a and b must be 16-byte aligned");
  Fp8Args g{out, static_cast<const uint8_t*>(a), static_cast<const uint8_t*>(b), a_scales,
            a_scales_numel, b_scales, b_scales_numel, bias, m, n, k, lda, ldb, ldc, workspace,
            workspace_elems, static_cast<hipStream_t>(stream)};
  return MI355X_DISPATCH_HALF(out_dtype, [&] { return run_fp8<scalar_t, OpI8>(g); });
}
