// prefill_attention.hip — paged prefill / chunked-prefill attention (varlen, causal
// bottom-right aligned, GQA) over the SAME x-split paged KV cache the decode kernel reads.
//
// Reference call site: flash_attn_varlen_func(q, k=key_cache, v=value_cache,
// block_table=..., cu_seqlens_q, seqused_k, causal=True) in
// vllm_metax/v1/attention/backends/flash_attn.py:725-747 — the arithmetic lives in a closed
// third-party wheel; semantics follow the reference's own oracle for that boundary,
// tests/kernels/attention/test_flash_attn.py:27-80 (fp32 scores and softmax, probabilities
// rounded to the value dtype, fp32 PV accumulation).
//
// MI355X design (fast path: head_size 128, block_size 16, 16-bit types):
//   * one 256-thread workgroup = 128 consecutive query tokens of one (sequence, q-head);
//     each wave owns 32 query rows (2 MFMA column tiles);
//   * the x-split layout is MFMA-shaped by construction: a 16-token K block
//     [d/8][16][8] is the A operand of S^T = K.Q^T (lane (lr,lc) <- chunk 4*ds+lr, token
//     lc = 16 contiguous bytes), and a V block [d][16] is the A operand of O^T = V^T.P^T
//     (lane <- row 16*dt+lc, tokens 4*lr..4*lr+3 = 8 contiguous bytes).  K/V blocks are
//     copied verbatim global -> LDS (32 tokens per stage, double buffered) and every
//     fragment read is lane-linear (bank-conflict free);
//   * computing S^T (keys on rows, queries on columns) leaves each lane with 4+4 keys of ONE
//     query column, which after exp() and rounding IS the B operand of the PV MFMA (k index
//     = 8*lr + j <-> key (block j>>2, 4*lr + (j&3))): probabilities never leave registers;
//   * online softmax: every accumulator register of a lane belongs to the same query column,
//     so the rescale factor is one scalar per lane and column tile.
//   * fp8 (e4m3fn) KV cache (KV8): blocks are 2 KiB ([d/16][16][16] bytes / [d][16] bytes), copied
//     verbatim as well; a lane's MFMA fragment is 8 bytes of K (ds_read_b64) or 4 + 4 bytes of V,
//     converted exactly to scalar_t in registers (v_cvt_scalef32_pk_*_fp8, scale 1); k_scale is
//     folded into the softmax scale and v_scale into the final 1/l.
#include "common.cuh"

namespace mi355x {

template <typename T>
struct MfmaQK;
template <>
struct MfmaQK<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    bf16x2_t v = {static_cast<bf16_t>(lo), static_cast<bf16_t>(hi)};
    return __builtin_bit_cast(uint32_t, v);
  }
  // 4 e4m3 (E5M2: e5m2) bytes -> 4 bf16 (two packed words), exact
  template <bool E5M2 = false>
  static __device__ __forceinline__ uint2 from_fp8x4(uint32_t w) {
    if constexpr (E5M2)
      return make_uint2(
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, 1.0f, false)),
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, 1.0f, true)));
    else
      return make_uint2(
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false)),
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true)));
  }
};
template <>
struct MfmaQK<f16_t> {
  static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
    f16x2_t v = {static_cast<f16_t>(lo), static_cast<f16_t>(hi)};
    return __builtin_bit_cast(uint32_t, v);
  }
  template <bool E5M2 = false>
  static __device__ __forceinline__ uint2 from_fp8x4(uint32_t w) {
    if constexpr (E5M2)
      return make_uint2(
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w, 1.0f, false)),
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w, 1.0f, true)));
    else
      return make_uint2(
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false)),
          __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true)));
  }
};

// all-reduce over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) in registers: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange row pairs / wave halves (vdst's odd rows with src's even rows;
// vdst's upper half with src's lower half), so op(swap(x, x)) is the xor-16 / xor-32 butterfly step without the
// LDS round trip of ds_bpermute (__shfl_xor).
__device__ __forceinline__ float rows_max(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
  const uint32_t v = __builtin_bit_cast(uint32_t, x);
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)q[0]), __builtin_bit_cast(float, (uint32_t)q[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
  const uint32_t v = __builtin_bit_cast(uint32_t, x);
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __builtin_bit_cast(float, (uint32_t)q[0]) + __builtin_bit_cast(float, (uint32_t)q[1]);
}

constexpr int kPfD = 128;
constexpr int kPfBS = 16;
constexpr int kPfQTile = 128;   // query rows per workgroup
constexpr int kPfKvTile = 32;   // keys per stage (2 cache blocks)
constexpr int kPfStages = 3;    // LDS ring depth (16 KiB per stage)
constexpr float kNegBig = -1.0e30f;

// IMG: `out` is the activation operand image of the prefill GEMM that consumes the attention output (o_proj):
// [row tile of 16 tokens][k tile of 32][64 slots of 8 elements], slot lr * 16 + (lc ^ g(lr)) = O[16 mt + lc][32 kt +
// 8 lr ..] over the [tokens, num_heads * 128] matrix — the re-tiling launch in front of that GEMM disappears, and
// the epilogue's stores become two 256-byte runs per instruction instead of sixteen 32-byte ones.
// OPTS: the optional arguments of the call site (flash_attn.py:725-747) — window_size = (W - 1, 0), softcap,
// alibi_slopes — on the same MFMA kernel: the window skips the key stages below the workgroup's first visible key
// and masks per row, the cap costs one tanh per score (cap * tanh(s / cap), before the mask:
// tests/kernels/attention/test_flash_attn.py:60-67), ALiBi adds slope[head] * (key - query position) to the
// scaled score (the bias of the decode kernel, attention_kernels.cuh:286, at every query position).  A separate
// instantiation: the plain causal path keeps its register count.
// QT: 16-row query tiles per wave — 2 (128 query rows per workgroup) or 1 (64: twice the workgroups for grids that
// leave the chip half empty — chunked prefill of one or two sequences, a TP shard's few heads; same bits per row).
template <typename T, bool KV8, bool IMG = false, bool OPTS = false, bool E5M2 = false, int QT = 2>
__global__ __launch_bounds__(256, 2) void paged_prefill_d128_kernel(
    T* __restrict__ out, const T* __restrict__ q, const void* __restrict__ k_cache_v,
    const void* __restrict__ v_cache_v, int num_heads, int num_kv_heads, float scale,
    const int* __restrict__ block_tables, const int* __restrict__ seq_lens,
    const int* __restrict__ cu_seqlens_q, int max_num_blocks_per_seq, int q_blocks_per_seq,
    int64_t q_stride, int64_t out_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    const float* __restrict__ k_scale, const float* __restrict__ v_scale,
    const int64_t* __restrict__ positions = nullptr, const T* __restrict__ cos_sin_cache = nullptr,
    int window = 0, float softcap = 0.f, const float* __restrict__ alibi_slopes = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [stage][K: 2 blocks | V: 2 blocks], a block = 4 KiB (scalar_t cache) or 2 KiB (fp8 cache)
  uint4* lds = reinterpret_cast<uint4*>(smem);
  constexpr int kBlkBytes = KV8 ? 2048 : 4096;
  constexpr int kStageVec = 4 * kBlkBytes / 16;  // uint4 per stage (16 / 8 KiB)
  constexpr int kCopies = KV8 ? 2 : 4;           // 1-KiB DMA pieces per wave and stage
  const char* k_cache = static_cast<const char*>(k_cache_v);   // strides below are in cache elements
  const char* v_cache = static_cast<const char*>(v_cache_v);
  constexpr int kCe = KV8 ? 1 : (int)sizeof(T);                // bytes per cache element
  if constexpr (KV8) scale *= *k_scale;

  // heaviest query blocks first: under the causal mask block qb needs (qb + 1) * 4 key stages, and
  // workgroups are dispatched in blockIdx order — the long ones must not be the last to start
  // (1-D grid: block = (reversed query block, head, sequence), sequence fastest)
  const int num_seqs_g = (int)gridDim.x / (q_blocks_per_seq * num_heads);
  const int per_qb = num_heads * num_seqs_g;
  const int qb = q_blocks_per_seq - 1 - (int)blockIdx.x / per_qb;
  const int rem = (int)blockIdx.x - (q_blocks_per_seq - 1 - qb) * per_qb;
  const int head = rem / num_seqs_g;
  const int seq = rem - head * num_seqs_g;
  const int q_begin = cu_seqlens_q[seq];
  const int q_len = cu_seqlens_q[seq + 1] - q_begin;
  constexpr int kQTile = 64 * QT;   // query rows per workgroup
  const int m0 = qb * kQTile;
  if (m0 >= q_len) return;
  const int seq_len = seq_lens[seq];
  const int ctx = seq_len - q_len;
  const int kv_head = head / (num_heads / num_kv_heads);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15;
  const int lr = lane >> 4;

  // ---- Q^T B-operand fragments: qf[qt][ds] = Q[row][32*ds + 8*lr .. +7] ----------------
  uint4 qf[QT][4];
  int qrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = m0 + wave * (16 * QT) + qt * 16 + lc;
    const int r = qrow[qt] < q_len ? qrow[qt] : q_len - 1;
    const T* qp = q + (int64_t)(q_begin + r) * q_stride + (int64_t)head * kPfD + 8 * lr;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[qt][ds] = *reinterpret_cast<const uint4*>(qp + 32 * ds);
    if (positions != nullptr) {
      // NeoX rotary of the query rows on the fly (rot_dim = 128): element d pairs with d + 64, i.e. fragment ds
      // with ds + 2 of the same lane; cos | sin of the row's position (the arithmetic of rotary_embedding:
      // csrc/pos_encoding_kernels.cu:10-34) — the q half of the rotary launch in front of this kernel
      const T* cs_row = cos_sin_cache + positions[q_begin + r] * kPfD + 8 * lr;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        Vec16<T> x, y;
        *reinterpret_cast<uint4*>(x.e) = qf[qt][ds];
        *reinterpret_cast<uint4*>(y.e) = qf[qt][ds + 2];
        const Vec16<T> cs = load16(cs_row + 32 * ds);
        const Vec16<T> sn = load16(cs_row + 64 + 32 * ds);
#pragma unroll
        for (int j = 0; j < 8; ++j) rot_pair<T>(x.e[j], y.e[j], cs.e[j], sn.e[j]);
        qf[qt][ds] = *reinterpret_cast<const uint4*>(x.e);
        qf[qt][ds + 2] = *reinterpret_cast<const uint4*>(y.e);
      }
    }
  }

  f32x4_t oacc[QT][8];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) oacc[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  float mrun[QT], lrun[QT];    // (lrun: per-lane partial sums, its own keys only)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = kNegBig;
    lrun[qt] = 0.f;
  }

  // keys needed by this workgroup: [0, kv_end)
  const int kv_end = min(seq_len, ctx + m0 + kQTile);
  const int num_tiles = (kv_end + kPfKvTile - 1) / kPfKvTile;
  const int* block_table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  const int num_seq_blocks = (seq_len + kPfBS - 1) / kPfBS;

  // staging: LDS-DMA, no registers (a register-staged prefetch was spilled to scratch by the
  // compiler with a full vmcnt(0) wait per 16-B load: 4.7 us per stage, profiles/r01_*notes*).
  // Stage image = [K blk0 | K blk1 | V blk0 | V blk1], 4 KiB each, copied verbatim; wave w copies
  // bytes [1024 w, 1024 w + 1024) of each block: 4 lane-linear copies per wave and stage, into a
  // ring of kPfStages stages (two stages in flight behind the math).
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  // block ids of the sequence: 64 at a time in one VGPR, read back with readlane
  int bt_reg = 0;
  int bt_base = -64;
  auto block_id = [&](int blk) {
    if (blk >= bt_base + 64 || blk < bt_base) {
      bt_base = blk & ~63;
      const int i = bt_base + lane;
      bt_reg = block_table[i < num_seq_blocks ? i : num_seq_blocks - 1];
    }
    return (int64_t)__builtin_amdgcn_readlane(bt_reg, blk - bt_base);
  };
  auto stage_issue = [&](int slot, int tile) {
    if constexpr (!KV8) {
      // wave w copies bytes [1024 w, 1024 w + 1024) of each of the four 4-KiB blocks
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int blk = tile * 2 + (i & 1);
        blk = blk < num_seq_blocks ? blk : num_seq_blocks - 1;
        const int64_t pb = block_id(blk);
        const char* base = (i < 2 ? k_cache : v_cache) +
                           (pb * kv_block_stride + (int64_t)kv_head * kv_head_stride) * kCe;
        lds_dma16(base + wave * 1024 + lane * 16,
                  lds_base + slot * (kStageVec * 16) + i * 4096 + wave * 1024);
      }
    } else {
      // eight 1-KiB pieces [K0 lo, K0 hi, K1 lo, K1 hi, V0 lo, ...]; wave w copies pieces w and w + 4
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = wave + 4 * i;
        const int b = piece >> 1;                       // 0: K blk0, 1: K blk1, 2: V blk0, 3: V blk1
        int blk = tile * 2 + (b & 1);
        blk = blk < num_seq_blocks ? blk : num_seq_blocks - 1;
        const int64_t pb = block_id(blk);
        const char* base = (b < 2 ? k_cache : v_cache) +
                           (pb * kv_block_stride + (int64_t)kv_head * kv_head_stride) * kCe;
        lds_dma16(base + (piece & 1) * 1024 + lane * 16,
                  lds_base + slot * (kStageVec * 16) + piece * 1024);
      }
    }
  };

  // sliding window: no row of this workgroup sees a key below (first query position) - W + 1
  int tile0 = 0;
  if constexpr (OPTS) {
    if (window > 0) tile0 = max(ctx + m0 - window + 1, 0) / kPfKvTile;
  }
  stage_issue(0, tile0);
  if (tile0 + 1 < num_tiles) stage_issue(1, tile0 + 1);

  // softmax scale in log2 units; with OPTS the scores are brought to scaled units first (cap / bias act there)
  const float sl2 = (OPTS ? 1.0f : scale) * 1.4426950408889634f;
  float slope = 0.f, cap_inv2 = 0.f;
  int wave_first = 0;   // first key any row of this wave may see
  if constexpr (OPTS) {
    if (alibi_slopes != nullptr) slope = alibi_slopes[head];
    if (softcap > 0.f) cap_inv2 = 2.0f * 1.4426950408889634f / softcap;   // tanh(x / cap) via 2^(2 x log2e / cap)
    if (window > 0) wave_first = ctx + m0 + wave * (16 * QT) - window + 1;
  }
  // every key <= this index is visible to EVERY query row of the wave (and is a real key)
  const int wave_limit_lo = min(ctx + m0 + wave * (16 * QT), seq_len - 1);
  // the wave's first query row decides which tiles it can skip entirely (causal)
  const int wave_q_hi = ctx + m0 + wave * (16 * QT) + 16 * QT - 1;  // last key any row of this wave may see

  int cur = 0;   // ring slot of `tile`
  for (int tile = tile0; tile < num_tiles; ++tile) {
    // stage `tile` has landed once only the copies of the next stage are pending
    if (tile + 1 < num_tiles) lds_dma_wait<kCopies>();
    else lds_dma_wait<0>();
    __syncthreads();   // everybody's share of stage `tile` is in LDS; stage tile-1 is dead
    if (tile + 2 < num_tiles) {
      int slot = cur + 2;
      slot = slot >= kPfStages ? slot - kPfStages : slot;
      stage_issue(slot, tile + 2);
    }

    const int t0 = tile * kPfKvTile;
    if (t0 + kPfKvTile > seq_len) {
      // last stage of the sequence: zero the V entries of keys >= seq_len once, in LDS (their probabilities are
      // exactly 0, and 0 * NaN must not reach O) — masking them inside the PV loop put a full lgkmcnt(0) and the
      // zeroing code behind every fragment read.  Thread -> row tid & 127 of V block tid >> 7.
      const int bi = tid >> 7;
      const int kvalid = seq_len - (t0 + 16 * bi);          // keys [0, kvalid) of this block are real
      char* vimg = reinterpret_cast<char*>(lds + cur * kStageVec) + 2 * kBlkBytes + bi * kBlkBytes;
      if constexpr (!KV8) {
        uint32_t* vrow = reinterpret_cast<uint32_t*>(vimg + (tid & 127) * 32);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          if (2 * w >= kvalid) vrow[w] = 0u;
          else if (2 * w + 1 >= kvalid) vrow[w] &= 0xFFFFu;
        }
      } else {
        uint32_t* vrow = reinterpret_cast<uint32_t*>(vimg + (tid & 127) * 16);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int live = kvalid - 4 * w;                    // live bytes of this dword
          if (live <= 0) vrow[w] = 0u;
          else if (live < 4) vrow[w] &= (1u << (8 * live)) - 1u;
        }
      }
      __syncthreads();
    }
    if (t0 <= wave_q_hi && (!OPTS || t0 + kPfKvTile > wave_first)) {
      const uint4* kbuf = lds + cur * kStageVec;
      const uint2* vbuf = reinterpret_cast<const uint2*>(lds + cur * kStageVec + 512);
      const char* sbuf = reinterpret_cast<const char*>(lds + cur * kStageVec);   // fp8 stage image
      // ---- S^T tiles: s[b][qt] = K_b . Q_qt^T, rows = keys 16*b + 4*lr + j, col = query lc
      f32x4_t s[2][QT];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[b][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          uint4 kf;
          if constexpr (!KV8) {
            kf = kbuf[b * 256 + ds * 64 + lane];
          } else {
            // d = 32 ds + 8 lr + j -> piece 2 ds + (lr >> 1), bytes 8 (lr & 1) .. +7 of token lc
            const uint2 raw = *reinterpret_cast<const uint2*>(
                sbuf + b * 2048 + ((2 * ds + (lr >> 1)) * 16 + lc) * 16 + 8 * (lr & 1));
            const uint2 lo = MfmaQK<T>::template from_fp8x4<E5M2>(raw.x), hi = MfmaQK<T>::template from_fp8x4<E5M2>(raw.y);
            kf = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[b][qt] = MfmaQK<T>::run(kf, qf[qt][ds], s[b][qt]);
        }
      }
      // ---- online softmax per query column ------------------------------------------
      // Scores stay in raw (unscaled) units; the softmax scale and log2(e) are folded into one
      // fma in front of v_exp_f32 (2^x): p = 2^(s*c - m*c), c = scale*log2(e).  Masking costs VALU
      // only on tiles that reach past the wave's first visible-key limit (the diagonal / tail);
      // the accumulator rescale is skipped when no lane's running maximum moved (alpha == 1).
      uint4 pfrag[QT];
      const bool need_mask = OPTS || (t0 + kPfKvTile - 1) > wave_limit_lo;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float v[8];
        [[maybe_unused]] unsigned vis = 0xFFu;   // OPTS: which of the lane's 8 keys this row sees
        if constexpr (OPTS) {
          const int qpos = ctx + qrow[qt];
          const int limit = min(qpos, seq_len - 1);
          const int first = window > 0 ? qpos - window + 1 : 0;
          vis = 0u;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int key = t0 + 16 * b + 4 * lr + j;
              float sv = s[b][qt][j] * scale;
              if (softcap > 0.f) {
                // cap * tanh(sv / cap), tanh(x) = 1 - 2 / (e^(2x) + 1)
                const float e = __builtin_amdgcn_exp2f(sv * cap_inv2);
                sv = softcap * (1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f));
              }
              sv = fmaf(slope, (float)(key - qpos), sv);
              const bool ok = key <= limit && key >= first;
              vis |= ok ? (1u << (b * 4 + j)) : 0u;
              v[b * 4 + j] = ok ? sv : kNegBig;
            }
          }
        } else if (need_mask) {
          const int limit = min(ctx + qrow[qt], seq_len - 1);  // last visible key of this row
#pragma unroll
          for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int key = t0 + 16 * b + 4 * lr + j;
              v[b * 4 + j] = key <= limit ? s[b][qt][j] : kNegBig;   // (also drops NaN of dead keys)
            }
          }
        } else {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[b * 4 + j] = s[b][qt][j];
          }
        }
        float tmax = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])),
                           fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
        tmax = rows_max(tmax);
        const float mnew = fmaxf(mrun[qt], tmax);
        const float alpha = __builtin_amdgcn_exp2f((mrun[qt] - mnew) * sl2);
        mrun[qt] = mnew;
        const float mc = -mnew * sl2;
        float p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(fmaf(v[i], sl2, mc));
        if constexpr (OPTS) {
          // a row whose window has not started yet has NO visible key in this stage: its running maximum is still
          // the sentinel and 2^(sentinel - sentinel) would count every masked key as 1
#pragma unroll
          for (int i = 0; i < 8; ++i) p[i] = (vis >> i) & 1u ? p[i] : 0.f;
        }
        const float psum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        lrun[qt] = lrun[qt] * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) {
            oacc[qt][dt][0] *= alpha;
            oacc[qt][dt][1] *= alpha;
            oacc[qt][dt][2] *= alpha;
            oacc[qt][dt][3] *= alpha;
          }
        }
        pfrag[qt].x = MfmaQK<T>::pack(p[0], p[1]);
        pfrag[qt].y = MfmaQK<T>::pack(p[2], p[3]);
        pfrag[qt].z = MfmaQK<T>::pack(p[4], p[5]);
        pfrag[qt].w = MfmaQK<T>::pack(p[6], p[7]);
      }
      // ---- O^T += V^T . P^T ------------------------------------------------------------
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        // V block image [128 d][16 keys]: row 16*dt + lc, keys 4*lr..4*lr+3 -> 8 bytes
        uint2 v0, v1;
        if constexpr (!KV8) {
          v0 = vbuf[(0 * 4096 + (16 * dt + lc) * 32 + 8 * lr) / 8];
          v1 = vbuf[(1 * 4096 + (16 * dt + lc) * 32 + 8 * lr) / 8];
        } else {   // V block image [128 d][16 keys] bytes: row 16 dt + lc, keys 4 lr .. 4 lr + 3
          v0 = MfmaQK<T>::template from_fp8x4<E5M2>(*reinterpret_cast<const uint32_t*>(
              sbuf + 4096 + (16 * dt + lc) * 16 + 4 * lr));
          v1 = MfmaQK<T>::template from_fp8x4<E5M2>(*reinterpret_cast<const uint32_t*>(
              sbuf + 6144 + (16 * dt + lc) * 16 + 4 * lr));
        }
        const uint4 vf = make_uint4(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = MfmaQK<T>::run(vf, pfrag[qt], oacc[qt][dt]);
      }
    }
    cur = cur + 1 == kPfStages ? 0 : cur + 1;
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ----------------------------------------------
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lrun[qt];
    l = rows_sum(l);
    float inv = 1.0f / l;
    if constexpr (KV8) inv *= *v_scale;
    if (qrow[qt] < q_len) {
      if constexpr (IMG) {
        // element (token t, column head * 128 + 16 dt + 4 lr + j): row tile t >> 4, row t & 15; k tile
        // head * 4 + (dt >> 1), k group 2 (dt & 1) + (lr >> 1), half (lr & 1) of that 16-byte slot
        const int t = q_begin + qrow[qt];
        const int kt32 = num_heads * (kPfD / 32);
        uint2* img = reinterpret_cast<uint2*>(out) + ((int64_t)(t >> 4) * kt32 + head * 4) * 128 + (lr & 1);
        const int lcp = t & 15;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const uint2 v = make_uint2(MfmaQK<T>::pack(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                                     MfmaQK<T>::pack(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
          const int lrp = 2 * (dt & 1) + (lr >> 1);
          const int slot = lrp * 16 + (lcp ^ (((lrp & 1) * 12) | (lrp & 2)));
          img[((dt >> 1) * 64 + slot) * 2] = v;
        }
      } else {
      T* op = out + (int64_t)(q_begin + qrow[qt]) * out_stride + (int64_t)head * kPfD + 4 * lr;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const uint2 v = make_uint2(MfmaQK<T>::pack(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv),
                                   MfmaQK<T>::pack(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv));
        *reinterpret_cast<uint2*>(op + 16 * dt) = v;
      }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Generic fallback (any head size / block size / dtype): one wave per (query token, head),
// lanes stride over keys, two passes (max+sum, then PV).  Correctness path only.
template <typename CT>
__device__ __forceinline__ float cache_to_f32(CT v) {
  if constexpr (std::is_same<CT, e5m2_t>::value) return bf8_to_f32(v.v);
  else if constexpr (sizeof(CT) == 1) return fp8_to_f32((uint8_t)v);
  else return to_f32(v);
}

template <typename T, typename CT>
__global__ __launch_bounds__(64) void paged_prefill_generic_kernel(
    T* __restrict__ out, const T* __restrict__ q, const CT* __restrict__ k_cache,
    const CT* __restrict__ v_cache, int num_heads, int num_kv_heads, int head_size,
    int block_size, float scale, const int* __restrict__ block_tables,
    const int* __restrict__ seq_lens, const int* __restrict__ cu_seqlens_q, int num_seqs,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t out_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, const float* __restrict__ k_scale, const float* __restrict__ v_scale,
    int window, float softcap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);  // [head_size]
  const int tok = blockIdx.x;
  const int head = blockIdx.y;
  const int lane = threadIdx.x;
  constexpr int X = 16 / sizeof(CT);
  float vsc = 1.f;
  if constexpr (sizeof(CT) == 1 && sizeof(T) != 1) {
    scale *= *k_scale;
    vsc = *v_scale;
  }
  // locate the sequence of this token (num_seqs is small; linear scan)
  int seq = 0;
  while (seq + 1 < num_seqs && cu_seqlens_q[seq + 1] <= tok) ++seq;
  if (tok >= cu_seqlens_q[num_seqs]) return;
  const int q_begin = cu_seqlens_q[seq];
  const int q_len = cu_seqlens_q[seq + 1] - q_begin;
  const int seq_len = seq_lens[seq];
  const int visible = seq_len - q_len + (tok - q_begin) + 1;  // keys [0, visible)
  const int kv_head = head / (num_heads / num_kv_heads);
  const int* block_table = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  for (int d = lane; d < head_size; d += 64) {
    qs[d] = to_f32(q[(int64_t)tok * q_stride + (int64_t)head * head_size + d]);
  }
  __syncthreads();
  auto kdot = [&](int key) {
    const int64_t pb = block_table[key / block_size];
    const int off = key % block_size;
    const CT* kp = k_cache + pb * kv_block_stride + (int64_t)kv_head * kv_head_stride;
    float acc = 0.f;
    for (int d = 0; d < head_size; ++d) {
      acc += qs[d] * cache_to_f32<CT>(kp[((d / X) * block_size + off) * X + (d % X)]);
    }
    acc *= scale;
    // logit soft-capping (the call site's `softcap`, flash_attn.py:725-747): cap * tanh(s / cap), on
    // the scaled score, before masking — as the reference's test oracle test_flash_attn.py:66-67
    if (softcap > 0.f) acc = softcap * tanhf(acc / softcap);
    return acc;
  };
  // sliding window (`window_size = (W - 1, 0)` at the call site): a query at absolute position p sees
  // keys p - W + 1 .. p (test_flash_attn.py:60-65)
  const int first = window > 0 ? max(visible - window, 0) : 0;
  float m = kNegBig;
  for (int key = first + lane; key < visible; key += 64) m = fmaxf(m, kdot(key));
  m = wave_max(m);
  float lsum = 0.f;
  for (int key = first + lane; key < visible; key += 64) lsum += __expf(kdot(key) - m);
  lsum = wave_sum(lsum);
  const float inv = 1.0f / lsum;
  // PV: lanes over d, serial over keys (probabilities recomputed; rounded to T like the oracle)
  for (int d = lane; d < head_size; d += 64) {
    float acc = 0.f;
    for (int key = first; key < visible; ++key) {
      const int64_t pb = block_table[key / block_size];
      const int off = key % block_size;
      const CT* vp = v_cache + pb * kv_block_stride + (int64_t)kv_head * kv_head_stride;
      const float p = to_f32(from_f32<T>(__expf(kdot(key) - m) * inv));
      acc += p * cache_to_f32<CT>(vp[(int64_t)d * block_size + off]);
    }
    out[(int64_t)tok * out_stride + (int64_t)head * head_size + d] = from_f32<T>(acc * vsc);
  }
}

}  // namespace mi355x

using namespace mi355x;

static int paged_prefill_impl(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, const int* cu_seqlens_q, int max_query_len,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t out_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_cache_dtype, const float* k_scale,
    const float* v_scale, int sliding_window, float softcap, mi355x_stream stream, bool image,
    const int64_t* positions = nullptr, const void* cos_sin_cache = nullptr,
    const float* alibi_slopes = nullptr) {
  MI355X_REQUIRE(sliding_window >= 0 && softcap >= 0.f, MI355X_EINVAL,
                 "paged_prefill_attention: sliding_window / softcap must be >= 0 (0 = off)");
  MI355X_REQUIRE(kv_cache_dtype == MI355X_KV_AUTO || kv_cache_dtype == MI355X_KV_FP8_E4M3 ||
                     kv_cache_dtype == MI355X_KV_FP8_E5M2,
                 MI355X_EUNSUPPORTED, "Unsupported data type of kv cache: id %d", kv_cache_dtype);
  const bool kv8 = kv_cache_dtype != MI355X_KV_AUTO;
  const bool e5m2 = kv_cache_dtype == MI355X_KV_FP8_E5M2;
  MI355X_REQUIRE(!kv8 || (k_scale && v_scale), MI355X_EINVAL,
                 "paged_prefill_attention: fp8 KV cache needs k_scale / v_scale");
  MI355X_REQUIRE(!kv8 || head_size % 16 == 0, MI355X_EUNSUPPORTED,
                 "paged_prefill_attention: fp8 KV cache needs head_size %% 16 == 0");
  MI355X_REQUIRE(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && head_size > 0 &&
                     block_size > 0 && max_query_len >= 0,
                 MI355X_EINVAL, "paged_prefill_attention: bad sizes");
  MI355X_REQUIRE(num_heads % num_kv_heads == 0, MI355X_EINVAL,
                 "paged_prefill_attention: num_heads %% num_kv_heads != 0");
  if (num_seqs == 0 || max_query_len == 0) return MI355X_OK;
  MI355X_REQUIRE(out && query && key_cache && value_cache && block_tables && seq_lens &&
                     cu_seqlens_q,
                 MI355X_EINVAL, "paged_prefill_attention: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // sliding window / soft-cap / ALiBi run on the OPTS instantiation of the MFMA kernel
  const bool opts = sliding_window > 0 || softcap > 0.f || alibi_slopes != nullptr;
  const bool fast = head_size == 128 && block_size == 16 && dtype != MI355X_F32 &&
                    q_stride % 8 == 0 && out_stride % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(query) | reinterpret_cast<uintptr_t>(key_cache) |
                      reinterpret_cast<uintptr_t>(value_cache)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 7) == 0 &&
                    kv_block_stride % (kv8 ? 16 : 8) == 0 && kv_head_stride % (kv8 ? 16 : 8) == 0;
  if (image && !(fast && !opts && (reinterpret_cast<uintptr_t>(out) & 15) == 0)) return 1;   // image form: plain fast path only
  MI355X_REQUIRE(fast || alibi_slopes == nullptr, MI355X_EUNSUPPORTED,
                 "paged_prefill_attention: alibi_slopes need head_size 128, block_size 16 and a 16-bit dtype");
  if (fast) {
    // 128 query rows per workgroup — 64 when that leaves fewer than two workgroups per CU (chunked prefill of one or
    // two sequences, a TP shard's few heads: a workgroup is a serial chain over its key stages, and 128 workgroups take
    // as long as 512; MI355X_PF_QT=1|2 forces a shape for A/B runs)
    static const int forced_qt = [] { const char* e = getenv("MI355X_PF_QT"); return e ? atoi(e) : 0; }();
    const int q_blocks128 = (max_query_len + kPfQTile - 1) / kPfQTile;
    const bool qt1 = forced_qt ? forced_qt == 1 : (int64_t)num_seqs * q_blocks128 * num_heads < 512;
    const int q_tile = qt1 ? 64 : kPfQTile;
    const int q_blocks = (max_query_len + q_tile - 1) / q_tile;
    dim3 grid(num_seqs * q_blocks * num_heads), block(256);
#define PF_KERN(...) (qt1 ? paged_prefill_d128_kernel<scalar_t, __VA_ARGS__, 1> : paged_prefill_d128_kernel<scalar_t, __VA_ARGS__, 2>)
    const size_t smem = (size_t)kPfStages * 4 * (kv8 ? 2048 : 4096);   // ring of 16- / 8-KiB stages
    return MI355X_DISPATCH_HALF(dtype, [&] {
      // (a function pointer takes no default arguments: all 23 spelled out)
      auto launch = [&](auto kern, bool rope, bool with_opts) {
        hipLaunchKernelGGL(kern, grid, block, smem, s, static_cast<scalar_t*>(out),
                           static_cast<const scalar_t*>(query), key_cache, value_cache, num_heads, num_kv_heads, scale,
                           block_tables, seq_lens, cu_seqlens_q, max_num_blocks_per_seq, q_blocks, q_stride,
                           out_stride, kv_block_stride, kv_head_stride, k_scale, v_scale,
                           rope ? positions : nullptr,
                           rope ? static_cast<const scalar_t*>(cos_sin_cache) : nullptr,
                           with_opts ? sliding_window : 0, with_opts ? softcap : 0.f,
                           with_opts ? alibi_slopes : nullptr);
      };
      if (image) {
        if (kv8 && e5m2) launch(PF_KERN(true, true, false, true), true, false);
        else if (kv8) launch(PF_KERN(true, true, false, false), true, false);
        else launch(PF_KERN(false, true, false, false), true, false);
        return check_launch("paged_prefill_attention_image");
      }
      if (opts) {
        if (kv8 && e5m2) launch(PF_KERN(true, false, true, true), false, true);
        else if (kv8) launch(PF_KERN(true, false, true, false), false, true);
        else launch(PF_KERN(false, false, true, false), false, true);
        return check_launch("paged_prefill_attention(opts)");
      }
      if (kv8 && e5m2) launch(PF_KERN(true, false, false, true), false, false);
      else if (kv8) launch(PF_KERN(true, false, false, false), false, false);
      else launch(PF_KERN(false, false, false, false), false, false);
      return check_launch("paged_prefill_attention");
    });
#undef PF_KERN
  }
  // generic path needs the total number of query tokens: upper bound num_seqs * max_query_len
  const int64_t max_tokens = (int64_t)num_seqs * max_query_len;
  MI355X_REQUIRE(max_tokens <= 0x7fffffff, MI355X_EUNSUPPORTED, "paged_prefill_attention: too many tokens");
  dim3 grid((int)max_tokens, num_heads), block(64);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    if (kv8 && e5m2) {
      hipLaunchKernelGGL((paged_prefill_generic_kernel<scalar_t, e5m2_t>), grid, block,
                         (size_t)head_size * sizeof(float), s, static_cast<scalar_t*>(out),
                         static_cast<const scalar_t*>(query), static_cast<const e5m2_t*>(key_cache),
                         static_cast<const e5m2_t*>(value_cache), num_heads, num_kv_heads, head_size,
                         block_size, scale, block_tables, seq_lens, cu_seqlens_q, num_seqs,
                         max_num_blocks_per_seq, q_stride, out_stride, kv_block_stride,
                         kv_head_stride, k_scale, v_scale, sliding_window, softcap);
    } else if (kv8) {
      hipLaunchKernelGGL((paged_prefill_generic_kernel<scalar_t, uint8_t>), grid, block,
                         (size_t)head_size * sizeof(float), s, static_cast<scalar_t*>(out),
                         static_cast<const scalar_t*>(query), static_cast<const uint8_t*>(key_cache),
                         static_cast<const uint8_t*>(value_cache), num_heads, num_kv_heads, head_size,
                         block_size, scale, block_tables, seq_lens, cu_seqlens_q, num_seqs,
                         max_num_blocks_per_seq, q_stride, out_stride, kv_block_stride,
                         kv_head_stride, k_scale, v_scale, sliding_window, softcap);
    } else {
      hipLaunchKernelGGL((paged_prefill_generic_kernel<scalar_t, scalar_t>), grid, block,
                         (size_t)head_size * sizeof(float), s, static_cast<scalar_t*>(out),
                         static_cast<const scalar_t*>(query), static_cast<const scalar_t*>(key_cache),
                         static_cast<const scalar_t*>(value_cache), num_heads, num_kv_heads, head_size,
                         block_size, scale, block_tables, seq_lens, cu_seqlens_q, num_seqs,
                         max_num_blocks_per_seq, q_stride, out_stride, kv_block_stride,
                         kv_head_stride, k_scale, v_scale, sliding_window, softcap);
    }
    return check_launch("paged_prefill_attention(generic)");
  });
}

extern "C" int mi355x_paged_prefill_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, const int* cu_seqlens_q, int max_query_len,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t out_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_cache_dtype, const float* k_scale,
    const float* v_scale, int sliding_window, float softcap, mi355x_stream stream) {
  return paged_prefill_impl(out, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size,
                            block_size, scale, block_tables, seq_lens, cu_seqlens_q, max_query_len,
                            max_num_blocks_per_seq, q_stride, out_stride, kv_block_stride, kv_head_stride, dtype,
                            kv_cache_dtype, k_scale, v_scale, sliding_window, softcap, stream, false);
}

extern "C" int mi355x_paged_prefill_attention_alibi(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, const int* cu_seqlens_q, int max_query_len,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t out_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_cache_dtype, const float* k_scale,
    const float* v_scale, int sliding_window, float softcap, const float* alibi_slopes,
    mi355x_stream stream) {
  return paged_prefill_impl(out, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size,
                            block_size, scale, block_tables, seq_lens, cu_seqlens_q, max_query_len,
                            max_num_blocks_per_seq, q_stride, out_stride, kv_block_stride, kv_head_stride, dtype,
                            kv_cache_dtype, k_scale, v_scale, sliding_window, softcap, stream, false, nullptr,
                            nullptr, alibi_slopes);
}

// returns 1 (no error) when the image form does not apply: run mi355x_paged_prefill_attention instead
extern "C" int mi355x_paged_prefill_attention_image(
    void* image, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, const int* cu_seqlens_q, int max_query_len,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_cache_dtype, const float* k_scale, const float* v_scale, const int64_t* positions,
    const void* cos_sin_cache, mi355x_stream stream) {
  MI355X_REQUIRE((positions == nullptr) == (cos_sin_cache == nullptr), MI355X_EINVAL,
                 "paged_prefill_attention_image: positions and cos_sin_cache go together");
  MI355X_REQUIRE((reinterpret_cast<uintptr_t>(cos_sin_cache) & 15) == 0, MI355X_EINVAL,
                 "paged_prefill_attention_image: cos_sin_cache must be 16-byte aligned");
  return paged_prefill_impl(image, query, key_cache, value_cache, num_seqs, num_heads, num_kv_heads, head_size,
                            block_size, scale, block_tables, seq_lens, cu_seqlens_q, max_query_len,
                            max_num_blocks_per_seq, q_stride, /*out_stride (unused)*/ 4, kv_block_stride,
                            kv_head_stride, dtype, kv_cache_dtype, k_scale, v_scale, 0, 0.f, stream, true, positions,
                            cos_sin_cache);
}
