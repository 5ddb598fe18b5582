// w4a16_unfused.hip — prefill-sized w4a16 GEMM as two kernels: (1) dequantise the int4 weight
// matrix ONCE into an MFMA-operand-shaped bf16/f16 workspace, (2) a 256x256x64 MFMA GEMM whose
// inner loop is pure LDS-read + MFMA (no dequant VALU competing with the matrix pipe).
//
// Why: measured on MI355X (profiles/r01_pmc_w4a16_large.txt) the fused kernel issues 4.8 VALU
// instructions per MFMA (exact dequant costs ~3 VALU ops per weight and is replicated per
// 128-row tile), so its VALU issue time (30 % of the kernel) exceeds its MFMA time (25 %).
// Dequantising once costs N*K*2.5 bytes of extra HBM traffic (< 5 % of the GEMM time at
// M = 8192) and makes the contraction MFMA-bound.  The reference takes the same route for
// shapes its fused kernel does not cover (awq.py:136-139: awq_dequantize + matmul).
//
// Numerics are those of the fused kernels (w4a16.cuh): w = T(fma(q, s, -z*s)) exactly as the
// reference (hgemm_gptq.h:487-570, 869-905), fp32 MFMA accumulation, one rounding of C.
//
// Packed-B workspace layout (produced by kernel 1, consumed by kernel 2): pieces of 1 KiB
//   P[nt = N/16][kt = K/32][slot 0..63][8 x T]
// piece (nt, kt) is the LDS image of one 16-column x 32-k MFMA B operand: the 8 consecutive-k
// values of column-lane lc, k-group lr live at slot swz(lr, lc) = lr*16 + (lc ^ g[lr]),
// g = {0, 12, 2, 14}.  That XOR makes BOTH the fragment read (lane (lr,lc) -> ds_read_b128,
// lane groups {0-3,12-15,20-27}, ...) AND the row-major staging write (lane l -> row l/4,
// chunk l%4 -> ds_write_b128, 8-lane groups) bank-conflict free (derivation in DESIGN.md §3).
// Column tiles are interleaved inside a 64-column group exactly as in the fused kernel
// (tile t owns columns 64q + 4c + t), so a lane's 4 accumulators are 4 consecutive columns.
#include <cstdlib>

#include "w4a16.cuh"

namespace mi355x {

// ---------------------------------------------------------------- kernel 1: dequant + pack
// one thread per (column, 32-k step): 4 shuffled words (4 x 8 consecutive k of the column) -> the
// 4 slots the column owns in one piece.  A 256-thread workgroup reads 4 packed rows x 1 KiB
// (contiguous) and writes 16 complete 1-KiB pieces; scale and zero point are fetched once per
// thread (group sizes are multiples of 32).
template <typename T, int ZMODE>
__global__ __launch_bounds__(256) void w4_dequant_pack_kernel(
    T* __restrict__ packed, const uint32_t* __restrict__ qw, const T* __restrict__ scales,
    const uint32_t* __restrict__ qz, int n, int k, int group) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kt32 = k >> 5;
  if (idx >= (int64_t)kt32 * n) return;
  const int kt = (int)(idx / n);
  const int col = (int)(idx - (int64_t)kt * n);
  const int g = (kt * 32) / group;
  uint32_t w[4];
#pragma unroll
  for (int lr = 0; lr < 4; ++lr) w[lr] = qw[(int64_t)(kt * 4 + lr) * n + col];
  const float s = to_f32(scales[(int64_t)g * n + col]);
  const uint32_t zw = qz[(int64_t)g * (n >> 3) + (col >> 3)];
  float z;
  if constexpr (ZMODE == kZeroAwq) {
    z = (float)((zw >> (4 * awq_shift(col & 7))) & 0xFu);
  } else {
    z = (float)(((zw >> (4 * (col & 7))) & 0xFu) + 1u);
  }
  const float zs = -z * s;
  // destination piece / slots
  const int q = col >> 6, c = (col & 63) >> 2, t = col & 3;
  const int nt = q * 4 + t;
  uint4* piece = reinterpret_cast<uint4*>(packed) + ((int64_t)nt * kt32 + kt) * 64;
#pragma unroll
  for (int lr = 0; lr < 4; ++lr) piece[frag_swz(lr, c)] = dequant_word<T>(w[lr], s, zs);
}

// 8-bit GPTQ weights (words [K/4][N], 4 consecutive k of one column per word, zero = qzeros + 1):
// same image, one thread per (column, 32-k step) reads 8 words.
template <typename T>
__global__ __launch_bounds__(256) void w8_dequant_pack_kernel(
    T* __restrict__ packed, const uint32_t* __restrict__ qw, const T* __restrict__ scales,
    const uint32_t* __restrict__ qz, int n, int k, int group) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kt32 = k >> 5;
  if (idx >= (int64_t)kt32 * n) return;
  const int kt = (int)(idx / n);
  const int col = (int)(idx - (int64_t)kt * n);
  const int g = (kt * 32) / group;
  const float s = to_f32(scales[(int64_t)g * n + col]);
  const float z = (float)(((qz[(int64_t)g * (n >> 2) + (col >> 2)] >> (8 * (col & 3))) & 0xFFu) + 1u);
  const float zs = -z * s;
  const int q = col >> 6, c = (col & 63) >> 2, t = col & 3;
  const int nt = q * 4 + t;
  uint4* piece = reinterpret_cast<uint4*>(packed) + ((int64_t)nt * kt32 + kt) * 64;
#pragma unroll
  for (int lr = 0; lr < 4; ++lr) {
    const uint32_t w0 = qw[(int64_t)(kt * 8 + 2 * lr) * n + col];
    const uint32_t w1 = qw[(int64_t)(kt * 8 + 2 * lr + 1) * n + col];
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = fmaf((float)((w0 >> (8 * j)) & 0xFFu), s, zs);
      f[4 + j] = fmaf((float)((w1 >> (8 * j)) & 0xFFu), s, zs);
    }
    piece[frag_swz(lr, c)] = make_uint4(Mfma<T>::pack(f[0], f[1]), Mfma<T>::pack(f[2], f[3]),
                                        Mfma<T>::pack(f[4], f[5]), Mfma<T>::pack(f[6], f[7]));
  }
}

// (kernel 2, pack_a_kernel: activations -> operand images, lives in w4a16.cuh)

// ---------------------------------------------------------------- kernel 3: the GEMM
// C[M,N] = PA . PB^T with BOTH operands pre-packed as MFMA operand images: every staging
// transfer is a lane-linear 1 KiB LDS-DMA copy (no VGPR staging, no address shuffles), the
// inner loop is ds_read_b128 + MFMA only.  256 x 256 tile, 8 waves (2 x 4), each wave 128 x 64.
// K advances in stages of 32 (32 KiB of LDS: 16 A pieces + 16 B pieces); kUfStages stages form
// a ring, the copies of stage t + kUfStages - 1 are issued while stage t is consumed, so up to
// three stages (96 KiB per CU) are in flight behind the matrix pipe.  The two waves of a SIMD run
// the loop half an iteration apart (ping-pong, see the loop): one is in its MFMA cluster while the
// other issues its copies and fragment reads.  M = 8192, kernel alone (scripts/bench_gemm_pp.py):
// 1358 -> 1493 TFLOP/s on random operands, 1626 -> 2000 on all-zero ones (what is left on random
// data is the board's clock under toggling operands, not the schedule).
constexpr int kUfBM = 256;
constexpr int kUfBN = 256;
constexpr int kUfBK = 32;
// (Round-2 ablation arms — copy placement orders 0 / 2, lgkmcnt after the barrier, shorter issue distance, the
//  in-phase round-1 loop, the MFMA-less fill-rate build — were measured in profiles/r02_gemm_pingpong_ab.txt and
//  removed from the source in round 3; what is kept is the one schedule that won.)
constexpr int kUfStages = 4;
constexpr int kUfDist = kUfStages - 1;       // stages between a copy's issue and its first read
constexpr int kUfThreads = 512;

// SILU = true: the GEMM is a gate_up projection (n = 2 * ffn, gate columns first) and the epilogue
// writes act[M, ffn] = silu_and_mul(C) instead of C (csrc/activation_kernels.cu:14-36): a tile is
// then 128 gate columns + the 128 matching up columns, and every wave owns 2 gate column tiles
// and the 2 matching up column tiles, so gate and up of one output element sit in ONE lane.
// MODE: 0 = plain C; 1 = SILU epilogue, act row-major; 2 = SILU epilogue, act written as the packed
// operand image of the NEXT GEMM (down_proj): the tile's 256 x 128 results are exchanged through
// LDS so that every wave stores whole 1-KiB pieces, and the activation re-tiling launch of the next
// GEMM (a read + write of the largest activation of the layer) disappears.
template <typename T, int MODE>
__global__ __launch_bounds__(kUfThreads, 2) void gemm_packed_kernel(
    T* __restrict__ c, const uint4* __restrict__ pa, const uint4* __restrict__ pb, int m, int n,
    int k, int num_m_blocks, int num_tiles, float* __restrict__ slabs, int sk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  // per stage: A pieces [16 mt][64] then B pieces [16 nt][64]  (uint4 units)
  constexpr int kStage = 2 * 16 * 64;  // 2048 uint4 = 32 KiB
  constexpr int kBOff = 16 * 64;

  // sk > 1 (few tiles: chunked-prefill sized M, narrow N): workgroup (tile, split) runs one K range and leaves its
  // fp32 partial tile in slabs[split][m][n] (SILU: gate and up columns at their places in [m][n]); a finish kernel
  // adds the slabs in split order, rounds to T and — SILU — applies silu_and_mul on the T values, writing act
  // row-major or as the operand image: the bits of the plain split GEMM followed by the two ops (deterministic, no
  // atomics).  grid = num_tiles * sk, split-major.
  const int split = blockIdx.x / num_tiles;
  int tile;
  {
    const int b = blockIdx.x - split * num_tiles;
    const int q = num_tiles / 8, r = num_tiles % 8;
    const int xcd = b % 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
  }
  // grouped rasterisation: consecutive tiles sweep 8 row-blocks per column-block, so the ~32
  // tiles resident on one XCD form an 8 x 4 patch that shares A and B K-slices in its L2
  // (a 32 x 1 strip re-streams every A panel from HBM/MALL for every column block).
  int mb, nb;
  {
    constexpr int GM = 8;
    const int num_n_blocks = num_tiles / num_m_blocks;
    const int group = tile / (GM * num_n_blocks);
    const int first_m = group * GM;
    const int gsz = min(num_m_blocks - first_m, GM);
    const int within = tile - group * GM * num_n_blocks;
    mb = first_m + within % gsz;
    nb = within / gsz;
  }

  constexpr bool SILU = MODE != 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2;   // 0..1 : rows wm*128 ..
  const int wn = wave & 3;    // 0..3 : 64-column group
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int kt32 = k >> 5;
  const int per_split = (kt32 + sk - 1) / sk;
  const int kt0 = split * per_split;                              // first stage of this workgroup's K range
  const int ktiles = max(0, min(per_split, kt32 - kt0));

  // staging: wave w copies A pieces 2w, 2w+1 and B pieces 2w, 2w+1 of each stage
  // (piece = 16-row tile; its successive k-steps are adjacent in global memory)
  const uint4* a_src[2];
  const uint4* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave * 2 + i;
    int gmt = mb * 16 + p;
    const int max_mt = ((m + 15) >> 4) - 1;
    gmt = gmt < max_mt ? gmt : max_mt;
    a_src[i] = pa + (int64_t)gmt * kt32 * 64 + lane;
    int gnt = nb * 16 + p;
    if constexpr (SILU) gnt = p < 8 ? nb * 8 + p : (n >> 5) + nb * 8 + (p - 8);   // gate | up tiles
    const int max_nt = (n >> 4) - 1;
    gnt = gnt < max_nt ? gnt : max_nt;
    b_src[i] = pb + (int64_t)gnt * kt32 * 64 + lane;
  }
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave * 2 + i;
      lds_dma16(a_src[i] + (int64_t)(kt0 + kt) * 64, lds_base + (buf * kStage + p * 64) * 16);
      lds_dma16(b_src[i] + (int64_t)(kt0 + kt) * 64, lds_base + (buf * kStage + kBOff + p * 64) * 16);
    }
  };
  constexpr int kPerStage = 4;  // copies one wave issues per stage

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

#pragma unroll
  for (int s = 0; s < kUfDist; ++s) {
    if (s < ktiles) stage(s, s);
  }

  const int frag = frag_swz(lr, lc);  // this lane's slot inside a piece
  int cur = 0;                        // ring slot of stage kt
  // Two wave groups (waves 0-3 / 4-7: one of each per SIMD) run the same loop half an iteration
  // apart: an iteration is {memory cluster: copies of stage kt+S-1, the 12 fragment reads of stage
  // kt} barrier {32 MFMAs} barrier, and group 1 enters the loop one barrier late, so that while one
  // wave of a SIMD is in its MFMA cluster its partner is in its memory cluster.  Barrier epochs:
  // group 0 reads stage k in epoch 2k, group 1 in epoch 2k+1.
  //  * RAW: a wave retires its own copies of stage k+1 (counted vmcnt) at the end of its memory
  //    cluster of stage k (epochs 2k / 2k+1); the first read of stage k+1 is in epoch 2k+2.
  //  * WAR: slot (k-1) % S is re-staged in the memory cluster of stage k (epoch >= 2k); its last
  //    reads (group 1, epoch 2k-1) were retired by the lgkmcnt(0) in front of that epoch's barrier.
  if (ktiles >= kUfDist) lds_dma_wait<kPerStage * (kUfDist - 1)>();   // own copies of stage 0
  else lds_dma_wait<0>();
  __syncthreads();
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int nxt = kt + kUfDist;
    int slot = cur + kUfDist;
    slot = slot >= kUfStages ? slot - kUfStages : slot;
    const uint4* abuf = lds + cur * kStage + (wm * 8) * 64 + frag;
    const uint4* bbuf = lds + cur * kStage + kBOff + (SILU ? wn * 2 : wn * 4) * 64 + frag;
    uint4 bf[4], af[8];
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = bbuf[(SILU ? (t < 2 ? t : 6 + t) : t) * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = abuf[i * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (nxt < ktiles) stage(slot, nxt);
    if (nxt < ktiles) lds_dma_wait<kPerStage * (kUfDist - 1)>();   // own copies of stage kt+1
    else lds_dma_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[i][t] = Mfma<T>::run(af[i], bf[t], acc[i][t]);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur = cur + 1 == kUfStages ? 0 : cur + 1;
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();

  // ---- epilogue: lane holds 4 consecutive columns per (i, j) --------------------------------
  if constexpr (SILU) {
    if (sk > 1) {
      // tiles t = 0,1: gate columns col, col+1; t = 2,3: the same columns of the up half
      const int half = n >> 1;
      const int col = nb * 128 + 64 * (wn >> 1) + 4 * lc + 2 * (wn & 1);
      if (col >= half) return;
      float* sl = slabs + (int64_t)split * m * n;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = mb * kUfBM + wm * 128 + i * 16 + 4 * lr + j;
          if (row < m) {
            *reinterpret_cast<float2*>(sl + (int64_t)row * n + col) = make_float2(acc[i][0][j], acc[i][1][j]);
            *reinterpret_cast<float2*>(sl + (int64_t)row * n + half + col) = make_float2(acc[i][2][j], acc[i][3][j]);
          }
        }
      }
      return;
    }
  }
  if constexpr (MODE == 2) {
    // (n % 256 == 0: every tile is 128 whole act columns; rows >= m of the last row tile are zero
    // in the packed A, so their results are silu(0) * 0 = 0, as pack_a_kernel would write them)
    __syncthreads();            // every wave is done with the ring: reuse it as the exchange buffer
    const int cl = 64 * (wn >> 1) + 4 * lc + 2 * (wn & 1);     // first of this lane's 2 act columns
    const int chunk = (cl & 31) >> 3;                          // 16-B slot column inside the piece
    char* ex = smem + (cl >> 5) * 1024 + chunk * 256 + (cl & 7) * 2;
    const int swz_g = ((chunk & 1) * 12) | (chunk & 2);        // frag_swz's row xor
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rowl = wm * 128 + i * 16 + 4 * lr + j;
        const T g0 = from_f32<T>(acc[i][0][j]), g1 = from_f32<T>(acc[i][1][j]);
        const T u0 = from_f32<T>(acc[i][2][j]), u1 = from_f32<T>(acc[i][3][j]);
        T o[2] = {mul_t<T>(silu_t<T>(g0), u0), mul_t<T>(silu_t<T>(g1), u1)};
        // piece (rowl / 16) * 4 + cl / 32, slot frag_swz(chunk, rowl % 16)
        *reinterpret_cast<uint32_t*>(ex + (rowl >> 4) * 4096 + (((rowl & 15) ^ swz_g) << 4)) =
            *reinterpret_cast<const uint32_t*>(o);
      }
    }
    __syncthreads();
    const int kt_out = n >> 6;                      // 32-k pieces per row tile of act [m, n / 2]
    const int m_tiles = (m + 15) >> 4;
    uint4* dst = reinterpret_cast<uint4*>(c);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int piece = wave * 8 + q;
      const int gmt = mb * 16 + (piece >> 2);
      const int gkt = nb * 4 + (piece & 3);
      if (gmt < m_tiles) dst[((int64_t)gmt * kt_out + gkt) * 64 + lane] = lds[piece * 64 + lane];
    }
    return;
  }
  if constexpr (MODE == 1) {
    // tiles t = 0,1: gate columns col, col+1; t = 2,3: the same columns of the up half
    const int half = n >> 1;
    const int col = nb * 128 + 64 * (wn >> 1) + 4 * lc + 2 * (wn & 1);
    if (col >= half) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * kUfBM + wm * 128 + i * 16 + 4 * lr + j;
        if (row < m) {
          // exactly gemm -> T, then silu_and_mul on the T values
          const T g0 = from_f32<T>(acc[i][0][j]), g1 = from_f32<T>(acc[i][1][j]);
          const T u0 = from_f32<T>(acc[i][2][j]), u1 = from_f32<T>(acc[i][3][j]);
          T o[2] = {mul_t<T>(silu_t<T>(g0), u0), mul_t<T>(silu_t<T>(g1), u1)};
          *reinterpret_cast<uint32_t*>(c + (int64_t)row * half + col) = *reinterpret_cast<const uint32_t*>(o);
        }
      }
    }
    return;
  }
  const int ncol = nb * kUfBN + wn * 64 + 4 * lc;
  if (ncol >= n) return;
  if (sk > 1) {
    float* sl = slabs + (int64_t)split * m * n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mb * kUfBM + wm * 128 + i * 16 + 4 * lr + j;
        if (row < m)
          *reinterpret_cast<f32x4_t*>(sl + (int64_t)row * n + ncol) =
              f32x4_t{acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]};
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = mb * kUfBM + wm * 128 + i * 16 + 4 * lr + j;
      if (row < m) {
        const uint2 v = make_uint2(Mfma<T>::pack(acc[i][0][j], acc[i][1][j]),
                                   Mfma<T>::pack(acc[i][2][j], acc[i][3][j]));
        *reinterpret_cast<uint2*>(c + (int64_t)row * n + ncol) = v;
      }
    }
  }
}

// finish of a K-split gate_up GEMM: act = silu_and_mul(T(sum of slabs)) — one thread = 8 act columns of one row (16 B);
// IMAGE: written as the operand image of the next GEMM (rows >= m of the last 16-row tile zero, as pack_a_kernel
// leaves them), else row-major [m][half]
template <typename T, bool IMAGE>
__global__ __launch_bounds__(256) void w4a16_sum_slabs_silu_kernel(T* __restrict__ out, const float* __restrict__ slabs,
                                                                   int m, int n, int sk) {
  const int half = n >> 1, c8 = half >> 3;
  const int rows = IMAGE ? (m + 15) / 16 * 16 : m;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * c8) return;
  const int row = (int)(idx / c8), c = (int)(idx - (int64_t)row * c8) * 8;
  T o[8];
  if (row < m) {
    float g[8], u[8];
    sum_slabs<8>(slabs + (int64_t)row * n + c, sk, (int64_t)m * n, g);
    sum_slabs<8>(slabs + (int64_t)row * n + half + c, sk, (int64_t)m * n, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = mul_t<T>(silu_t<T>(from_f32<T>(g[e])), from_f32<T>(u[e]));
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(0.f);
  }
  if constexpr (IMAGE) {
    // piece (row / 16, c / 32), slot frag_swz((c % 32) / 8, row % 16)
    uint4* dst = reinterpret_cast<uint4*>(out) + ((int64_t)(row >> 4) * (half >> 5) + (c >> 5)) * 64 +
                 frag_swz((c & 31) >> 3, row & 15);
    *dst = *reinterpret_cast<const uint4*>(o);
  } else {
    *reinterpret_cast<uint4*>(out + (int64_t)row * half + c) = *reinterpret_cast<const uint4*>(o);
  }
}

// (Round 3 built the obvious alternative — 4 waves, one per SIMD, a 128 x 128 quadrant each: a third fewer LDS bytes
//  per flop, 256 accumulators per lane, the next stage's fragments read into a second register set — and measured it
//  against this kernel on one box (scripts/ab_r03.sh): 1322-1332 vs 1493-1495 TFLOP/s on random operands, 1485 vs
//  1990 on zeros.  Without a SIMD partner the 16 fragment reads and 8 copy issues of a stage are not hidden; placed
//  between the MFMA groups they pushed hipcc over the 512-register file (B fragments in AGPRs, 40 scratch
//  accesses).  Removed from the source; the two-waves-per-SIMD ping-pong stays.)
static inline int64_t unfused_scratch_bytes(int m, int n, int k) {
  const int64_t m_pad = ((int64_t)m + 15) / 16 * 16;
  return ((int64_t)n + m_pad) * k * 2;
}

template <typename T>
static int launch_dequant_pack(const GemmArgs& g, T* packed_b) {
  const int64_t words = (int64_t)(g.k / 32) * g.n;   // one thread per (column, 32-k step)
  if (g.bits == 8) {
    hipLaunchKernelGGL(w8_dequant_pack_kernel<T>, dim3((words + 255) / 256), dim3(256), 0, g.stream,
                       packed_b, g.qw, static_cast<const T*>(g.scales), g.qz, g.n, g.k, g.group);
    return check_launch("w8_dequant_pack");
  }
  if (g.zmode == kZeroAwq) {
    hipLaunchKernelGGL((w4_dequant_pack_kernel<T, kZeroAwq>), dim3((words + 255) / 256), dim3(256), 0,
                       g.stream, packed_b, g.qw, static_cast<const T*>(g.scales), g.qz, g.n, g.k,
                       g.group);
  } else {
    hipLaunchKernelGGL((w4_dequant_pack_kernel<T, kZeroGptq>), dim3((words + 255) / 256), dim3(256), 0,
                       g.stream, packed_b, g.qw, static_cast<const T*>(g.scales), g.qz, g.n, g.k,
                       g.group);
  }
  return check_launch("w4_dequant_pack");
}

// mi355x_w4a16_prepack: the weights' operand image, once (weight-load time)
int w4a16_prepack_dispatch(const GemmArgs& g, int dtype) {
  if (g.k % kUfBK != 0 || g.n % 64 != 0) return 1;
  if (dtype == MI355X_BF16) return launch_dequant_pack<bf16_t>(g, static_cast<bf16_t*>(g.c));
  if (dtype == MI355X_F16) return launch_dequant_pack<f16_t>(g, static_cast<f16_t*>(g.c));
  return 1;
}

template <typename T, int MODE>
static int run_unfused(const GemmArgs& g) {
  constexpr bool SILU = MODE != 0;
  // scratch: [weight image (absent when the caller holds a prepacked one)] [activation image]
  T* scratch = static_cast<T*>(g.dq_ws);
  const T* packed_b;
  T* packed_a_dst;
  int rc;
  if (g.b_image != nullptr) {
    packed_b = static_cast<const T*>(g.b_image);
    packed_a_dst = scratch;
  } else {
    rc = launch_dequant_pack<T>(g, scratch);
    if (rc) return rc;
    packed_b = scratch;
    packed_a_dst = scratch + (int64_t)g.n * g.k;
  }
  const T* packed_a = packed_a_dst;
  if (g.a_packed) {
    packed_a = static_cast<const T*>(g.a);   // the producer already wrote the operand image
  } else {
    const int m_tiles = (g.m + 15) / 16;
    hipLaunchKernelGGL(pack_a_kernel<T>, dim3((g.k + kPackK - 1) / kPackK, m_tiles), dim3(256), 0,
                       g.stream, packed_a_dst, static_cast<const T*>(g.a), g.m, g.k, g.lda);
    rc = check_launch("pack_a");
    if (rc) return rc;
  }
  const int num_m_blocks = (g.m + kUfBM - 1) / kUfBM;
  // SILU: a tile = 128 gate + 128 up columns, i.e. one block per 128 output columns
  const int num_n_blocks = SILU ? (g.n / 2 + 127) / 128 : (g.n + kUfBN - 1) / kUfBN;
  const int num_tiles = num_m_blocks * num_n_blocks;
  const size_t smem = (size_t)kUfStages * 2048 * sizeof(uint4);  // 128 KiB
  auto kern = gemm_packed_kernel<T, MODE>;
  static PerDeviceOnce attr_once;  // one per instantiation (T, MODE), one bit per device
  int dev;
  if (attr_once.need(&dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      set_error("gemm_packed: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
      return MI355X_EUNSUPPORTED;
    }
    attr_once.mark(dev);
  }
  // K split where the tiles leave most of the chip idle: slabs behind the operand images
  int sk = 1;
  float* slabs = nullptr;
  {
    static const int forced = [] { const char* e = getenv("MI355X_W4_PACKED_SK"); return e ? atoi(e) : 0; }();
    const int64_t img_bytes = ((g.b_image ? 0 : (int64_t)g.n * g.k * 2) +
                               (g.a_packed ? 0 : (((int64_t)g.m + 15) / 16 * 16) * g.k * 2) + 15) / 16 * 16;
    if (g.dq_ws != nullptr && g.dq_ws_bytes > img_bytes && (reinterpret_cast<uintptr_t>(g.dq_ws) & 15) == 0 &&
        g.n % 16 == 0 && (reinterpret_cast<uintptr_t>(g.c) & 15) == 0) {
      sk = plan_tile_split(g.m, g.n, g.k / 64, (g.dq_ws_bytes - img_bytes) / 4, forced);
      slabs = reinterpret_cast<float*>(static_cast<char*>(g.dq_ws) + img_bytes);
    }
  }
  hipLaunchKernelGGL(kern, dim3(num_tiles * sk), dim3(kUfThreads), smem, g.stream, static_cast<T*>(g.c),
                     reinterpret_cast<const uint4*>(packed_a), reinterpret_cast<const uint4*>(packed_b),
                     g.m, g.n, g.k, num_m_blocks, num_tiles, slabs, sk);
  rc = check_launch("gemm_packed");
  if (rc || sk == 1) return rc;
  if constexpr (SILU) {
    const int64_t rows = MODE == 2 ? ((int64_t)g.m + 15) / 16 * 16 : g.m;
    const int64_t threads = rows * (g.n / 16);
    hipLaunchKernelGGL((w4a16_sum_slabs_silu_kernel<T, MODE == 2>), dim3((threads + 255) / 256), dim3(256), 0, g.stream,
                       static_cast<T*>(g.c), slabs, g.m, g.n, sk);
  } else {
    const int64_t n4 = (int64_t)g.m * g.n / 4;
    hipLaunchKernelGGL(w4a16_sum_slabs_kernel<T>, dim3((n4 + 255) / 256), dim3(256), 0, g.stream,
                       static_cast<T*>(g.c), slabs, n4, sk);
  }
  return check_launch("gemm_packed(sum)");
}

// returns 1 when not applicable (caller uses the fused kernel)
int w4a16_gemm_unfused_dispatch(const GemmArgs& g, int dtype) {
  if (g.b_image != nullptr) {
    // prepacked weights: the scratch only holds the activation image (none when `a` is one already)
    const int64_t need = g.a_packed ? 0 : (((int64_t)g.m + 15) / 16 * 16) * g.k * 2;
    if (need > 0 && (g.dq_ws == nullptr || g.dq_ws_bytes < need)) return 1;
  } else {
    if (g.dq_ws == nullptr) return 1;
    if (g.dq_ws_bytes < unfused_scratch_bytes(g.m, g.n, g.k)) return 1;
  }
  if (g.k % kUfBK != 0 || g.n % 64 != 0) return 1;
  // too few 256-row tiles to fill 256 CUs below 1024 rows — unless the weights' image exists already (no per-call
  // dequantisation of the whole matrix) and the plain epilogue can split K (run_unfused): then from 384 rows
  // (Llama-3-8B layer at M = 576: 580 us of stripe passes against 365; profiles/r03_w4a16_mid_m_image.txt)
  if (g.m < (g.b_image != nullptr ? kW4PrepackedMinM : 1024)) return 1;
  if (g.fuse_silu) {
    if (g.n % 256 != 0) return 1;   // gate and up halves must each be whole 128-column blocks
    if (g.out_packed) {
      if (dtype == MI355X_BF16) return run_unfused<bf16_t, 2>(g);
      if (dtype == MI355X_F16) return run_unfused<f16_t, 2>(g);
      return 1;
    }
    if (dtype == MI355X_BF16) return run_unfused<bf16_t, 1>(g);
    if (dtype == MI355X_F16) return run_unfused<f16_t, 1>(g);
    return 1;
  }
  if (g.out_packed) return 1;
  if (dtype == MI355X_BF16) return run_unfused<bf16_t, 0>(g);
  if (dtype == MI355X_F16) return run_unfused<f16_t, 0>(g);
  return 1;
}

}  // namespace mi355x
