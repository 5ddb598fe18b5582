// w4a16_stripe.hip — the decode-shaped (M <= 64) w4a16 GEMM: C[M, N] = A[M, K] . dequant(W)[K, N]
//
// Replaces, for the shapes it accepts, the generic small-M kernel of w4a16.hip on the op surface
// of the reference's awq_gemm / gptq_gemm (csrc/quantization/awq/gemm_kernels.cu:1165-1221,
// csrc/quantization/gptq/q_gemm.cu:2354-2413) — same operands, same exact dequantisation
// T(fma(q, s, -z*s)), fp32 accumulation.
//
// Regime (M = 64, Llama-3-8B layer shapes; measurements: profiles/r01_decode_gemm_notes.md): every
// weight byte is read once from HBM, the activations (M x K, <= 1.8 MB) are re-read by every
// workgroup from its XCD's L2.  What bounds a CU is the dequant VALU work (~96 VALU per wave and
// k-step, the byte->float and float->bf16 converts at half rate) and its L2/HBM -> LDS fill rate.
//   * one workgroup owns a STRIPE of BN = 64*NW columns (NW = 2 or 4 "column waves") and a
//     contiguous K range; its 8 waves are NW column waves x KW = 8/NW K waves.  The column waves
//     share the activation bytes of a k-step through LDS, so A traffic per weight byte is 4/NW of
//     what a 64-column tile pays.
//   * weights, activations, scales and zero points all arrive by LDS-DMA into rings of stages of
//     128 k (asynchronous, no VGPR staging).  A wave's DMA queue completes in issue order, so the
//     weight ring is fed by waves 0-3 and the activation/scale ring by waves 4-7; each waits only
//     on its own copies.  One workgroup barrier per stage.
//   * activations are copied straight from the row-major [M, K] tensor: the LDS side of an
//     LDS-DMA is lane-linear, so the choice of WHICH 16 bytes lane l fetches builds a
//     bank-conflict-free MFMA operand layout on the fly (no pre-tiling pass).
//   * K is split across workgroups only as far as needed to put ~all CUs on the weight
//     stream; partial tiles go to fp32 slabs ws[split][M][N] with plain stores and one small
//     kernel adds the slabs in split order and rounds to T (deterministic, no atomics: float
//     atomics run at ~1.3 TB/s chip-wide, plain stores at ~6).
//   * -DSTRIPE_STAMP is the one compile-time switch left (in-kernel s_memtime stamps, scripts/stamp_stripe.py);
//     the ablation arms behind the numbers in profiles/r0[12]_decode_gemm_notes.md were removed in round 3.
#include <cstdio>
#include <cstdlib>

#include "w4a16.cuh"

namespace mi355x {

// -DSTRIPE_DQ=n: dequant experiments (scripts/build_variant.sh): 0 = v_cvt_f32_ubyteN per weight (default),
// 1 = v_cvt_pk_f32_fp8 per weight pair (bit-identical), 2 = TIMING ONLY, WRONG VALUES: the 7-op 0x4300 | q conversion
// with no scale / zero applied at all = a lower bound for any "scale after the MMA" scheme in this loop structure
#ifndef STRIPE_DQ
#define STRIPE_DQ 0
#endif
// ring depths in stages (-DSTRIPE_WD / -DSTRIPE_AD for experiments)
// -DSTRIPE_PRIO=1: s_setprio 1 for waves 4-7 before the loop; 2: s_setprio 1 around the dequant + MFMA cluster
#ifndef STRIPE_PRIO
#define STRIPE_PRIO 0
#endif
#ifndef STRIPE_WD
#define STRIPE_WD 3
#endif
#ifndef STRIPE_AD
#define STRIPE_AD 3
#endif
#if STRIPE_DQ == 1
#define STRIPE_DEQUANT(w, s, zs) dequant_word_fp8pk<T>(w, s, zs)
#elif STRIPE_DQ == 2
__device__ __forceinline__ uint4 magic_word_no_scale(uint32_t w, float s, float zs) {
  const uint32_t m = 0x000F000Fu, c = 0x43004300u + (__builtin_bit_cast(uint32_t, s) & 1u) + (__builtin_bit_cast(uint32_t, zs) & 1u);
  uint4 r;
  r.x = (w & m) | c;
  r.y = ((w >> 4) & m) | c;
  r.z = ((w >> 8) & m) | c;
  r.w = ((w >> 12) & m) | c;
  return r;
}
#define STRIPE_DEQUANT(w, s, zs) magic_word_no_scale(w, s, zs)
#else
#define STRIPE_DEQUANT(w, s, zs) dequant_word<T>(w, s, zs)
#endif

constexpr double STRIPE_T256 = 1.25;   // cost-model time of a 256-k stage relative to a 128-k stage (bk = 256 plans)

// WV = waves per workgroup: 8 (stages of 128 k) or 16 (stages of 256 k, 4 waves per SIMD).  A wave's own
// instruction stream — ~120 VALU at one issue per 6-8 cycles, ~100 SALU, the LDS and copy-issue latencies —
// takes ~2000 cycles per stage whatever its SIMD partner does, so two waves per SIMD leave the vector issue
// port ~40 % idle; four fill it (profiles/r02_decode_gemm_notes.md).
template <int MT, int NW, int SETS, int WV>
struct StripeCfg {
  static constexpr int THREADS = 64 * WV;
  static constexpr int BK = 16 * WV;             // k per stage
  static constexpr int KSTEPS = BK / 32;
  static constexpr int LOADERS = WV / 2;         // waves [0, LOADERS) feed the weight ring, the rest the activation ring
  static constexpr int BN = 64 * NW;
  static constexpr int KW = WV / NW;             // waves along K
  static constexpr int KS_PER_WAVE = KSTEPS / KW;   // k-steps (of 32) of one stage per wave
  static constexpr int W_ROW_BYTES = BN * 4;     // one packed row = 8 k of every column
  static constexpr int W_BYTES = (BK / 8) * W_ROW_BYTES;
  static constexpr int SC_LANES = 8 * NW;        // 16-B pieces of the BN scales (2 B each)
  static constexpr int Z_LANES = 2 * NW;         // 16-B pieces of the BN/8 zero words
  static constexpr int SET_BYTES = 16 * (SC_LANES + Z_LANES);
  static constexpr int A_HALF_BYTES = MT * 4096; // one 128-k half: MT x 4 operand images of 1 KiB
  static constexpr int A_IMG_BYTES = (BK / 128) * A_HALF_BYTES;
  static constexpr int A_BYTES = A_IMG_BYTES + SETS * SET_BYTES;  // A-ring slot: images + sets
  // ring depths: the weight stream comes from HBM (~2-3 us under load) and needs >= 48 KiB in
  // flight per CU; activations / scales come from L2 and need two stages ahead.
  static constexpr int W_DEPTH = STRIPE_WD;
  static constexpr int A_DEPTH = STRIPE_AD;
  static constexpr int WAVES_PER_SIMD = WV / 4;
  static constexpr int W_COPIES = NW;            // per weight-loader wave (waves 0-3) and stage
  static constexpr int A_COPIES = MT + 1;        // per activation-loader wave (4-7): images + one set
  static constexpr int A_RING = 0;
  static constexpr int W_RING = A_DEPTH * A_BYTES;
  static constexpr int RING_BYTES = W_RING + W_DEPTH * W_BYTES;
  static constexpr int RED_BYTES = (KW / 2) * NW * MT * 16 * 64 * 4;   // tree reduction: the upper half writes
  static constexpr int LDS_BYTES = RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES;
};

// streamed-once variant of lds_dma16 (nt: do not keep the line in L2 for somebody else)
__device__ __forceinline__ void lds_dma16_nt(const void* gptr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
               :
               : "v"(gptr), "s"(lds_base)
               : "memory");
}

// SILU = true (NW == 2, no K split): the GEMM is a gate_up projection (n = 2 * ffn, gate columns
// first); a stripe is then 64 gate columns (column wave 0) + the 64 matching up columns (column
// wave 1) and the epilogue writes act[M, ffn] = silu_and_mul(C) (csrc/activation_kernels.cu:14-36
// applied to the T-rounded accumulators: the bits of awq_gemm followed by silu_and_mul).
template <typename T, int MT, int NW, int ZMODE, int SETS, bool SILU = false, int WV = 8>
__global__ __launch_bounds__((StripeCfg<MT, NW, SETS, WV>::THREADS)) void w4a16_gemm_stripe_kernel(
    T* __restrict__ c, float* __restrict__ slabs, const T* __restrict__ a, int64_t lda,
    const uint32_t* __restrict__ qw, const T* __restrict__ scales, const uint32_t* __restrict__ qz,
    int m, int n, int k, int group, int stages_per_split) {
  using Cfg = StripeCfg<MT, NW, SETS, WV>;
  constexpr int kStBK = Cfg::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef STRIPE_STAMP
  const uint64_t st_entry = __builtin_readcyclecounter();
  uint64_t st_loop0 = 0, st_loop1 = 0;
#endif

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % NW;   // column wave: columns n0 + 64*wn ..
  const int wk = wave / NW;   // K wave: k-steps wk*KS_PER_WAVE .. of every stage
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int n0 = blockIdx.x * Cfg::BN;
  // stripe-local column (multiple of 4) -> global column
  auto gcol = [&](int local) {
    if constexpr (SILU) return (local < 64 ? 0 : (n >> 1) - 64) + (int)blockIdx.x * 64 + local;
    else return n0 + local;
  };
  const int total_stages = k / kStBK;
  const int s_begin = blockIdx.y * stages_per_split;
  const int s_end = min(s_begin + stages_per_split, total_stages);
  const int nst = s_end - s_begin;
  // Every workgroup sweeps its K range from a different starting stage (wrapping around).  All
  // workgroups read the SAME activation rows, a stage's 256-byte row segments are one row pitch
  // apart and so live in ONE L2 channel (256-B interleave, pitch a multiple of 4 KiB): in
  // lock-step the 28-32 workgroups of an XCD would all queue on that channel.  Workgroups b, b+8,
  // b+16.. share an XCD (round-robin dispatch), so consecutive b/8 get consecutive start stages.
  // (the start stage is a function of the 128-column block index modulo HALF the block count, so
  //  that the gate and the up half of a gate_up projection — and the SILU variant, whose stripes
  //  are 64 + 64 columns — accumulate every column in the same order: bit-identical outputs)
  const unsigned bq = SILU ? (blockIdx.x >> 1) : blockIdx.x;
  const unsigned period = SILU ? (unsigned)(n >> 8) : ((gridDim.x + 1) >> 1);
  const unsigned br = bq % (period ? period : 1u);
  const int rot = nst > 0 ? (int)(((br >> 3) + 4u * (br & 7) + 7u * blockIdx.y) % (unsigned)nst) : 0;
  auto stage_of = [&](int i) {
    int s = i + rot;
    s = s >= nst ? s - nst : s;
    return s_begin + s;
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

  // ---- loader roles: the LDS-DMA queue of a wave completes in issue order, so the deep weight
  // ring and the shallow activation ring are fed by different waves (each waits on its own).
  const bool w_loader = wave < Cfg::LOADERS;
  const int lw = wave % Cfg::LOADERS;
  // weights: piece p = 1 KiB = 4/NW packed rows of the stripe; loader lw copies pieces lw*NW ..
  uint32_t w_off[Cfg::W_COPIES];   // byte offset from qw + stage * 16 * n
  {
    constexpr int RPP = 4 / NW;          // rows per piece
    constexpr int LPR = 64 / RPP;        // lanes per row
#pragma unroll
    for (int j = 0; j < Cfg::W_COPIES; ++j) {
      const int p = lw * Cfg::W_COPIES + j;
      const int r = p * RPP + lane / LPR;
      int col = gcol(4 * (lane % LPR));
      col = col <= n - 4 ? col : n - 4;
      w_off[j] = (uint32_t)(((int64_t)r * n + col) * 4);
    }
  }
  // activations: loader lw copies the 4 k-step images of row tile i for i = 0..MT-1, k-step lw
  // Activations, no pre-tiling pass: a copy moves 4 rows x 256 contiguous bytes (the 128 k of the
  // stage) of row tile i, rows 4j .. 4j+3 (j = lw), into piece (i, j) of the stage.  The LDS side
  // of an LDS-DMA is lane-linear, so WHICH 16 bytes lane l fetches decides the layout:
  //   slot 16 r + 4 (ks ^ j) + (lr ^ r)   <-  A[16 i + 4 j + r][32 ks + 8 lr .. +7]
  // A lane quad still reads one 64-byte run and 16 lanes one 256-byte row segment (coalesced
  // like a plain copy); the two XORs make the 16 lanes of every ds_read_b128 lane group
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ..) hit 16 different bank quads when lane (lr, lc)
  // reads row lc, k-group lr of one k-step — the MFMA A operand, conflict-free.
  uint32_t a_off[MT];   // byte offset from a + stage * 128
  {
    // (stages of 256 k: loader lw copies row quad lw & 3 of the 128-k half lw >> 2)
    const int aj = lw & 3, ah = lw >> 2;
    const int r = lane >> 4, ks = ((lane >> 2) & 3) ^ aj, a_lr = (lane & 3) ^ r;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int row = 16 * i + 4 * aj + r;
      row = row < m ? row : m - 1;   // rows >= m only feed accumulator rows that are never stored
      a_off[i] = (uint32_t)(((int64_t)row * lda + 128 * ah + 32 * ks + 8 * a_lr) * sizeof(T));
    }
  }
  // scales + zero points of one set: lanes [0, SC_LANES) fetch scales, the next Z_LANES zeros
  const int my_set = lw % SETS;
  const bool sc_lane = lane < Cfg::SC_LANES + Cfg::Z_LANES;
  const char* sc_src;       // + g * sc_gstride per group
  int64_t sc_gstride;
  if (lane < Cfg::SC_LANES) {
    int col = gcol(8 * lane);
    col = col <= n - 8 ? col : n - 8;
    sc_src = reinterpret_cast<const char*>(scales + col);
    sc_gstride = (int64_t)n * sizeof(T);
  } else {
    int zw = gcol(32 * (lane - Cfg::SC_LANES)) >> 3;
    zw = zw <= (n >> 3) - 4 ? zw : (n >> 3) - 4;
    sc_src = reinterpret_cast<const char*>(qz + zw);
    sc_gstride = (int64_t)(n >> 3) * 4;
  }

  auto issue_w = [&](int slot, int stage) {
    const uint32_t sb = lds_base + Cfg::W_RING + slot * Cfg::W_BYTES + lw * Cfg::W_COPIES * 1024;
#pragma unroll
    for (int j = 0; j < Cfg::W_COPIES; ++j)
      lds_dma16_s_nt(qw + (int64_t)stage * (kStBK / 8) * n, w_off[j], sb + j * 1024);
  };
  auto issue_a = [&](int slot, int stage) {
    const uint32_t sb = lds_base + Cfg::A_RING + slot * Cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      lds_dma16_s(a + stage * kStBK, a_off[i], sb + (lw >> 2) * Cfg::A_HALF_BYTES + (i * 4 + (lw & 3)) * 1024);
    const int g = (stage * kStBK + my_set * (kStBK / SETS)) / group;
    if (sc_lane) lds_dma16(sc_src + (int64_t)g * sc_gstride, sb + Cfg::A_IMG_BYTES + my_set * Cfg::SET_BYTES);
  };

  f32x4_t acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  if (w_loader) {
#pragma unroll
    for (int s = 0; s < Cfg::W_DEPTH - 1; ++s) {
      if (s < nst) issue_w(s, stage_of(s));
    }
  } else {
#pragma unroll
    for (int s = 0; s < Cfg::A_DEPTH - 1; ++s) {
      if (s < nst) issue_a(s, stage_of(s));
    }
  }

  // byte offset of this lane's A fragment inside a row tile, without the k-step term (see above)
  const int a_rd = (lc >> 2) * 1024 + (lc & 3) * 256 + ((lr ^ (lc & 3)) << 4);
  const int colw = 64 * wn + 4 * lc;       // first of this lane's 4 columns inside the stripe
  const int ncol = gcol(colw);
  int cur_w = 0, cur_a = 0;                // ring slots of stage `it`
  // The two waves of a SIMD (w and w + 4) run the loop half an iteration apart, as in the prefill GEMM
  // (w4a16_unfused.hip): an iteration is {memory cluster: LDS reads of the stage's operands, copies of a
  // later stage, own copies of stage it+1 retired} barrier {dequant + MFMA on registers} barrier, and waves
  // 4-7 enter the loop one barrier late.  In phase, both waves of a SIMD sit in their LDS / copy-issue
  // latencies together and then contend for the vector issue port together.
  //  * RAW: stage it+1 is first read two barriers after the counted vmcnt that retires a wave's copies of it.
  //  * WAR: the slot of stage it-1 is re-staged in the memory cluster of stage it; its last reads (waves
  //    4-7, one epoch earlier) were retired by the lgkmcnt(0) in front of that epoch's barrier.
  if (w_loader) {
    if (nst >= Cfg::W_DEPTH - 1) lds_dma_wait<Cfg::W_COPIES * (Cfg::W_DEPTH - 2)>();
    else lds_dma_wait<0>();
  } else {
    if (nst >= Cfg::A_DEPTH - 1) lds_dma_wait<Cfg::A_COPIES * (Cfg::A_DEPTH - 2)>();
    else lds_dma_wait<0>();
  }
  __syncthreads();   // stage 0 is complete in LDS
  if (!w_loader) __builtin_amdgcn_s_barrier();
#ifdef STRIPE_STAMP
  st_loop0 = __builtin_readcyclecounter();
  uint64_t st_sum[4] = {0, 0, 0, 0};
#define ST_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define ST_T(x)
#endif
#if STRIPE_PRIO == 1
  if (!w_loader) __builtin_amdgcn_s_setprio(1);   // (experiment: static priority for the later-dispatched half)
#endif
  for (int it = 0; it < nst; ++it) {
    ST_T(st0);
    const char* ab = smem + Cfg::A_RING + cur_a * Cfg::A_BYTES;
    const char* wb = smem + Cfg::W_RING + cur_w * Cfg::W_BYTES;
    uint2 scq[Cfg::KS_PER_WAVE];
    uint32_t zq[Cfg::KS_PER_WAVE];
    uint4 wq[Cfg::KS_PER_WAVE];
    uint4 af[Cfg::KS_PER_WAVE][MT];
#pragma unroll
    for (int q = 0; q < Cfg::KS_PER_WAVE; ++q) {
      const int ks = wk * Cfg::KS_PER_WAVE + q;
      const int set = ks / (Cfg::KSTEPS / SETS);
      const char* setp = ab + Cfg::A_IMG_BYTES + set * Cfg::SET_BYTES;
      scq[q] = *reinterpret_cast<const uint2*>(setp + colw * 2);
      zq[q] = *reinterpret_cast<const uint32_t*>(setp + 16 * Cfg::SC_LANES + (colw >> 3) * 4);
      wq[q] = *reinterpret_cast<const uint4*>(wb + (4 * ks + lr) * Cfg::W_ROW_BYTES + colw * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[q][i] = *reinterpret_cast<const uint4*>(ab + (ks >> 2) * Cfg::A_HALF_BYTES + i * 4096 + a_rd +
                                                   (((ks & 3) ^ (lc >> 2)) << 6));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (w_loader) {
      const int nxt = it + Cfg::W_DEPTH - 1;   // goes into the slot stage it-1 vacated
      int slot = cur_w + Cfg::W_DEPTH - 1;
      slot = slot >= Cfg::W_DEPTH ? slot - Cfg::W_DEPTH : slot;
      if (nxt < nst) {
        issue_w(slot, stage_of(nxt));
        lds_dma_wait<Cfg::W_COPIES * (Cfg::W_DEPTH - 2)>();   // own copies of stage it+1
      } else {
        lds_dma_wait<0>();
      }
    } else {
      const int nxt = it + Cfg::A_DEPTH - 1;
      int slot = cur_a + Cfg::A_DEPTH - 1;
      slot = slot >= Cfg::A_DEPTH ? slot - Cfg::A_DEPTH : slot;
      if (nxt < nst) {
        issue_a(slot, stage_of(nxt));
        lds_dma_wait<Cfg::A_COPIES * (Cfg::A_DEPTH - 2)>();
      } else {
        lds_dma_wait<0>();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ST_T(st1);
    __builtin_amdgcn_s_barrier();
    ST_T(st2);
#if STRIPE_PRIO == 2
    __builtin_amdgcn_s_setprio(1);                 // (experiment: the computing wave of a SIMD pair first)
#endif
    __builtin_amdgcn_sched_barrier(0);
    // Software pipeline over the 4*KS_PER_WAVE packed words of the stage: the MFMAs of word j
    // are issued between the dequant VALU of word j+1 (a lone wave issues a VALU op every ~8
    // cycles and an MFMA occupies the matrix pipe for 16: back to back they do not overlap).
    constexpr int NWORDS = 4 * Cfg::KS_PER_WAVE;
    float scf[Cfg::KS_PER_WAVE][4], zsf[Cfg::KS_PER_WAVE][4];
#pragma unroll
    for (int q = 0; q < Cfg::KS_PER_WAVE; ++q) {
      T sct[4];
      *reinterpret_cast<uint2*>(sct) = scq[q];
      float zp[4];
      unpack_zeros4<ZMODE>(zq[q], ncol, zp);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        scf[q][t] = to_f32(sct[t]);
        zsf[q][t] = -zp[t] * scf[q][t];
#if STRIPE_DQ == 1
        scf[q][t] *= 512.f;   // dequant_word_fp8pk takes 512 s
#endif
      }
    }
    auto word_of = [&](int j) {
      const int q = j >> 2, t = j & 3;
      return t == 0 ? wq[q].x : (t == 1 ? wq[q].y : (t == 2 ? wq[q].z : wq[q].w));
    };
    uint4 bf_cur = STRIPE_DEQUANT(word_of(0), scf[0][0], zsf[0][0]);
#pragma unroll
    for (int j = 0; j < NWORDS; ++j) {
      const int q = j >> 2, t = j & 3;
      uint4 bf_nxt = bf_cur;
      if (j + 1 < NWORDS) bf_nxt = STRIPE_DEQUANT(word_of(j + 1), scf[(j + 1) >> 2][(j + 1) & 3], zsf[(j + 1) >> 2][(j + 1) & 3]);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][t] = Mfma<T>::run(af[q][i], bf_cur, acc[i][t]);
      if (j + 1 < NWORDS) {
        // interleave: MT groups of {1 MFMA, ceil(19/MT) VALU}
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (19 + MT - 1) / MT, 0);
        }
      }
      bf_cur = bf_nxt;
    }
    __builtin_amdgcn_sched_barrier(0);
    ST_T(st3);
#if STRIPE_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef STRIPE_STAMP
    {
      const uint64_t st4 = __builtin_readcyclecounter();
      st_sum[0] += st1 - st0; st_sum[1] += st2 - st1; st_sum[2] += st3 - st2; st_sum[3] += st4 - st3;
    }
#endif
    cur_w = cur_w + 1 == Cfg::W_DEPTH ? 0 : cur_w + 1;
    cur_a = cur_a + 1 == Cfg::A_DEPTH ? 0 : cur_a + 1;
  }
#ifdef STRIPE_STAMP
  st_loop1 = __builtin_readcyclecounter();
  if (lane == 0 && blockIdx.x < 8 && gridDim.y == 1 && slabs != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) slabs[(blockIdx.x * 16 + wave) * 8 + i] = (float)st_sum[i] / (float)(nst > 0 ? nst : 1);
    slabs[(blockIdx.x * 16 + wave) * 8 + 4] = (float)(st_loop0 - st_entry);
    slabs[(blockIdx.x * 16 + wave) * 8 + 5] = (float)(st_loop1 - st_loop0);
  }
#endif
  if (w_loader) __builtin_amdgcn_s_barrier();

  // ---- add the KW K-waves of each column wave through LDS (the rings are dead now): a binary tree, in
  // every round the upper half of the live K-waves hands its tiles to the lower half (fixed order) ----
  float* red = reinterpret_cast<float*>(smem);
  constexpr int kSlab = MT * 16 * 64;
#pragma unroll
  for (int half = Cfg::KW / 2; half >= 1; half >>= 1) {
    __syncthreads();   // the rings / the previous round's slabs are dead
    if (wk >= half && wk < 2 * half) {
      // slot (i, j) of a tile is lane-linear (a wave instruction moves 1 KiB of consecutive LDS: no bank
      // conflicts; [row][64 columns] put lanes lr = 0..3 one KiB apart, 4-way conflicts on every access)
      float4* dst = reinterpret_cast<float4*>(red + ((wk - half) * NW + wn) * kSlab) + lane;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dst[(i * 4 + j) * 64] = make_float4(acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]);
      }
    }
    __syncthreads();
    if (wk < half) {
      const float4* src = reinterpret_cast<const float4*>(red + (wk * NW + wn) * kSlab) + lane;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = src[(i * 4 + j) * 64];
          acc[i][0][j] += v.x;
          acc[i][1][j] += v.y;
          acc[i][2][j] += v.z;
          acc[i][3][j] += v.w;
        }
      }
    }
  }
#ifdef STRIPE_STAMP
  if (lane == 0 && blockIdx.x < 8 && gridDim.y == 1 && slabs != nullptr)
    slabs[(blockIdx.x * 16 + wave) * 8 + 6] = (float)(__builtin_readcyclecounter() - st_loop1);   // reduction tree
#endif
  if constexpr (SILU) {
    // column wave 1 (up) hands its tile to column wave 0 (gate) through LDS: same lane layout
    __syncthreads();                 // all partial sums have been read
    if (wk == 0 && wn == 1) {
      float4* dst = reinterpret_cast<float4*>(red) + lane;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dst[(i * 4 + j) * 64] = make_float4(acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]);
      }
    }
    __syncthreads();
    if (wk != 0 || wn != 0) return;
    const int half = n >> 1;
    const int ocol = (int)blockIdx.x * 64 + 4 * lc;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = i * 16 + 4 * lr + j;
        if (row >= m) continue;
        const float4 u = reinterpret_cast<const float4*>(red)[(i * 4 + j) * 64 + lane];
        const float uu[4] = {u.x, u.y, u.z, u.w};
        T o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          o[t] = mul_t<T>(silu_t<T>(from_f32<T>(acc[i][t][j])), from_f32<T>(uu[t]));
        *reinterpret_cast<uint2*>(c + (int64_t)row * half + ocol) = *reinterpret_cast<const uint2*>(o);
      }
    }
    return;
  }
  if (wk > 0 || ncol >= n) return;
  float* slab = slabs + (int64_t)blockIdx.y * m * n;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = i * 16 + 4 * lr + j;
      if (row >= m) continue;
      if (gridDim.y == 1) {
        const uint2 v = make_uint2(Mfma<T>::pack(acc[i][0][j], acc[i][1][j]),
                                   Mfma<T>::pack(acc[i][2][j], acc[i][3][j]));
        *reinterpret_cast<uint2*>(c + (int64_t)row * n + ncol) = v;
      } else {
        *reinterpret_cast<float4*>(slab + (int64_t)row * n + ncol) =
            make_float4(acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]);
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------
struct StripePlan {
  int nw, sk, steps;
  double est_us;
};

// Cost model (microseconds) behind the choice of stripe width and K split; constants from
// /opt/skills/guides/MI355X_MICROARCH.md (per-CU LDS fill ~70 GB/s from L2, HBM ~6.5 TB/s
// streamed, plain stores ~6 TB/s) and the kernel's ~100 VALU per wave and k-step.
static StripePlan plan_stripe(int rows, int n, int k, int64_t ws_elems, int bk) {
  const int kStBK = bk;
  if (const char* f = getenv("MI355X_STRIPE_FORCE")) {   // "nw,sk" (kernel experiments)
    int nw = 2, sk = 1;
    if (sscanf(f, "%d,%d", &nw, &sk) == 2 && (nw == 2 || nw == 4) && sk >= 1) {
      const int stages = k / kStBK;
      if (sk > stages) sk = stages;
      if (sk > 1 && (int64_t)sk * rows * n > ws_elems) sk = 1;
      const int steps = (stages + sk - 1) / sk;
      return StripePlan{nw, (stages + steps - 1) / steps, steps, 0.0};
    }
  }
  const int mt = (rows + 15) / 16;
  const int mtt = mt <= 1 ? 1 : (mt <= 2 ? 2 : (mt <= 4 ? 4 : 8));
  const int stages = k / kStBK;
  const double wbytes = (double)n * k / 2;
  StripePlan best{0, 1, stages, 1e30};
  for (int nw = 2; nw <= 4; nw += 2) {
    if (bk == 256 && nw != 2) continue;   // (16 waves: 2 column waves x 8 K waves only)
    if (mtt == 8 && nw != 2) continue;    // (128-row passes: 256-column stripes would need > 160 KiB of rings)
    const int bn = 64 * nw;
    const int stripes = (n + bn - 1) / bn;
    const double fill = (64.0 * bn + mtt * 4096.0) / 70e3;
    const double valu = nw * 0.17;
    // per stage of bk k: two waves per SIMD (bk 128) are bound by a wave's own instruction stream,
    // four (bk 256) by the SIMD's vector issue port
    const double stage_t = (fill > valu ? fill : valu) * (bk == 256 ? STRIPE_T256 : 1.0);
    const int max_sk = stages < 16 ? stages : 16;
    for (int sk = 1; sk <= max_sk; ++sk) {
      const int steps = (stages + sk - 1) / sk;
      const int sk_eff = (stages + steps - 1) / steps;
      if (sk_eff != sk) continue;
      if (sk > 1 && (int64_t)sk * rows * n > ws_elems) continue;
      const int wgs = stripes * sk;
      const int rounds = (wgs + 255) / 256;
      double t = 2.0 + rounds * steps * stage_t;
      const double hbm = wbytes / 6.5e6;
      if (t < hbm) t = hbm;
      if (sk > 1) t += 3.0 + 2.0 * ((double)sk * rows * n * 4) / 5e6;
      if (t < best.est_us) best = StripePlan{nw, sk, steps, t};
    }
  }
  return best;
}

template <typename T, int MT, int NW, int ZMODE, int SETS, bool SILU = false, int WV = 8>
static int launch_stripe_cfg(const GemmArgs& g, const T* a, T* c, int rows, const StripePlan& p) {
  using Cfg = StripeCfg<MT, NW, SETS, WV>;
  auto kern = w4a16_gemm_stripe_kernel<T, MT, NW, ZMODE, SETS, SILU, WV>;
  static PerDeviceOnce attr_once;  // one per instantiation, one bit per device
  int dev;
  if (attr_once.need(&dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("w4a16_gemm_stripe: cannot reserve %d B of LDS: %s", Cfg::LDS_BYTES,
                hipGetErrorString(e));
      return MI355X_EUNSUPPORTED;
    }
    attr_once.mark(dev);
  }
  const int stripes = SILU ? g.n / 128 : (g.n + Cfg::BN - 1) / Cfg::BN;
  hipLaunchKernelGGL(kern, dim3(stripes, p.sk), dim3(Cfg::THREADS), Cfg::LDS_BYTES, g.stream, c, g.ws,
                     a, g.lda, g.qw, static_cast<const T*>(g.scales), g.qz,
                     rows, g.n, g.k, g.group, p.steps);
  return check_launch("w4a16_gemm_stripe");
}

template <typename T, int MT, int ZMODE, int SETS>
static int launch_stripe_nw(const GemmArgs& g, const T* a, T* c, int rows, const StripePlan& p, int wv) {
  (void)wv;
  if (p.nw == 2) return launch_stripe_cfg<T, MT, 2, ZMODE, SETS>(g, a, c, rows, p);
  return launch_stripe_cfg<T, MT, 4, ZMODE, SETS>(g, a, c, rows, p);
}

template <typename T, int ZMODE, int SETS>
static int launch_stripe_mt(const GemmArgs& g, const T* a, T* c, int rows, const StripePlan& p, int wv) {
  const int mt = (rows + 15) / 16;
  if (mt <= 1) return launch_stripe_nw<T, 1, ZMODE, SETS>(g, a, c, rows, p, wv);
  if (mt <= 2) return launch_stripe_nw<T, 2, ZMODE, SETS>(g, a, c, rows, p, wv);
  if (mt <= 4) return launch_stripe_nw<T, 4, ZMODE, SETS>(g, a, c, rows, p, wv);
  // 65 .. 128 rows per pass (chunked-prefill / mixed-batch sizes, 64 < M < 1024): the dequantised fragment of a
  // wave feeds 8 row tiles — per 128-k stage 32 MFMAs (512 matrix-pipe cycles) beside the same ~100 dequant VALU,
  // so a 128-row pass costs about what a 64-row pass does
  return launch_stripe_nw<T, 8, ZMODE, SETS>(g, a, c, rows, p, wv);
}

// waves per workgroup: 8.  (A 16-wave variant — stages of 256 k, four waves per SIMD — reached the vector-issue-port
// bound in its loop, ~1300 cycles per 128 k against ~1560, and gave it back in its prologue and 8-tile K-wave
// reduction: bench.py 5993 vs 5994 tokens/s, profiles/r02_decode_gemm_notes.md; removed from the source in round 3.)
static int stripe_waves(int, int, int) { return 8; }
// scale / zero sets per stage of bk k; 0: not a layout this kernel handles
static int stripe_sets(int group, int bk) {
  if (group % bk == 0) return 1;
  if (group < 32 || bk % group != 0) return 0;
  const int s = bk / group;
  return s <= 4 ? s : 0;
}

// returns 1 when the shape / scratch is not one this path handles (caller falls back)
template <typename T>
static int run_stripe(const GemmArgs& g, int row0, int rows) {
  if (rows > 128 || g.k % 128 != 0 || g.n % 64 != 0 || g.n < 64) return 1;
  if ((reinterpret_cast<uintptr_t>(g.qw) & 15) || (reinterpret_cast<uintptr_t>(g.scales) & 15) ||
      (reinterpret_cast<uintptr_t>(g.qz) & 15))
    return 1;
  const T* a = static_cast<const T*>(g.a) + (int64_t)row0 * g.lda;
  T* c = static_cast<T*>(g.c) + (int64_t)row0 * g.n;
  int rc;
  StripePlan p = plan_stripe(rows, g.n, g.k, g.ws ? g.ws_elems : 0, 128);
  if (rows > 64 && g.fuse_silu) return 1;
  if (p.nw == 0) return 1;
  const int wv = p.nw == 2 ? stripe_waves(g.k, g.group, p.sk) : 8;
  const int bk = 16 * wv;
  if (wv == 16) p = p.sk == 1 ? StripePlan{2, 1, g.k / bk, 0.0} : plan_stripe(rows, g.n, g.k, g.ws ? g.ws_elems : 0, bk);
  const int sets = stripe_sets(g.group, bk);
  if (sets == 0) return 1;
#define STRIPE_Z(SETSV)                                                              \
  (g.zmode == kZeroAwq ? launch_stripe_mt<T, kZeroAwq, SETSV>(g, a, c, rows, p, wv)         \
                       : launch_stripe_mt<T, kZeroGptq, SETSV>(g, a, c, rows, p, wv))
  rc = sets == 1 ? STRIPE_Z(1) : (sets == 2 ? STRIPE_Z(2) : STRIPE_Z(4));
#undef STRIPE_Z
  if (rc) return rc;
  if (p.sk > 1 && g.defer_sk != nullptr && row0 == 0 && rows == g.m) {
    *g.defer_sk = p.sk;   // the consumer adds the slabs (mi355x_fused_add_rms_norm_slabs)
    return MI355X_OK;
  }
  if (p.sk > 1) {
    const int64_t n4 = (int64_t)rows * g.n / 4;
    hipLaunchKernelGGL(w4a16_sum_slabs_kernel<T>, dim3((n4 + 255) / 256), dim3(256), 0, g.stream, c,
                       g.ws, n4, p.sk);
    rc = check_launch("w4a16_sum_slabs");
  }
  return rc;
}

// gate_up + silu_and_mul in one launch (decode): AWQ zero points, no K split, 64+64 columns per
// stripe.  Returns 1 when not applicable.
template <typename T>
static int run_stripe_silu(const GemmArgs& g) {
  if (g.m > 64 || g.m < 1 || g.k % 128 != 0 || g.n % 128 != 0 || g.zmode != kZeroAwq) return 1;
  const int wv = stripe_waves(g.k, g.group, 1);
  const int bk = 16 * wv;
  const int sets = stripe_sets(g.group, bk);
  if (sets == 0) return 1;
  if ((reinterpret_cast<uintptr_t>(g.qw) & 15) || (reinterpret_cast<uintptr_t>(g.scales) & 15) ||
      (reinterpret_cast<uintptr_t>(g.qz) & 15) || (reinterpret_cast<uintptr_t>(g.c) & 7))
    return 1;
  const T* a = static_cast<const T*>(g.a);
  T* c = static_cast<T*>(g.c);
  const StripePlan p{2, 1, g.k / bk, 0.0};
  const int mt = (g.m + 15) / 16;
#define STRIPE_SW(MTV, SETSV) launch_stripe_cfg<T, MTV, 2, kZeroAwq, SETSV, true, 8>(g, a, c, g.m, p)
#define STRIPE_S(MTV) (sets == 1 ? STRIPE_SW(MTV, 1) : (sets == 2 ? STRIPE_SW(MTV, 2) : STRIPE_SW(MTV, 4)))
  if (mt <= 1) return STRIPE_S(1);
  if (mt <= 2) return STRIPE_S(2);
  return STRIPE_S(4);
#undef STRIPE_S
#undef STRIPE_SW
}

int w4a16_gemm_stripe_silu_dispatch(const GemmArgs& g, int dtype) {
  if (dtype == MI355X_BF16) return run_stripe_silu<bf16_t>(g);
  if (dtype == MI355X_F16) return run_stripe_silu<f16_t>(g);
  return 1;
}

int w4a16_gemm_stripe_dispatch(const GemmArgs& g, int dtype, int row0, int rows) {
  if (dtype == MI355X_BF16) return run_stripe<bf16_t>(g, row0, rows);
  if (dtype == MI355X_F16) return run_stripe<f16_t>(g, row0, rows);
  return 1;
}

}  // namespace mi355x
