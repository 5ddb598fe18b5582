// w4a16.hip — int4 weight-only path for gfx950: AWQ->exllama repack, GPTQ shuffle,
// AWQ dequantize, and the w4a16 dequant-GEMM on MFMA (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Reference semantics restated:
//   awq_to_gptq_4bit  csrc/quantization/awq/gemm_kernels.cu:127-184, 323-356
//   awq_dequantize    csrc/quantization/awq/gemm_kernels.cu:96-125, 358-402
//   awq_gemm          csrc/quantization/awq/gemm_kernels.cu:410-463 -> hgemm_gptq.h
//                     :2165-2259; dequant w = T(fma(q, s, (-z)*s)) (hgemm_gptq.h:487-570,
//                     :869-905), fp32 MMA accumulation, one rounding of C to T
//   gptq_shuffle      csrc/quantization/gptq/q_gemm.cu:2321-2368, qdq_4.cuh:16-29,
//                     q_gemm.cu:2145-2174 (make_sequential)
//   gptq_gemm         csrc/quantization/gptq/q_gemm.cu:2373-2413 (zero = qzeros + 1, the
//                     generic exllama semantics :247-250; equals the fast path's fixed 8 on
//                     symmetric checkpoints), act-order gather perm_a :1770-1786
//
// Weight layout consumed by the GEMM (produced by awq_to_gptq_4bit / gptq_shuffle):
//   words [K/8][N]; nibble p of word (kk, n) = W[8kk + {0,2,4,6,1,3,5,7}[p]][n].
// One 32-bit word is exactly one lane's B fragment (8 consecutive k of one column) of
// v_mfma_f32_16x16x32: lane (r = l>>4, c = l&15) loads the 4 words (kk = 4*step + r,
// n = n0 + 4c .. 4c+3) with ONE 16-byte load, i.e. the fragments of 4 N-tiles whose
// columns are interleaved (tile t owns columns n0 + 4c + t).  No LDS staging for B.
//
// Dequant (exact): (q - z) * s is representable in fp32 for 4-bit q, z and a T scale, so
// T(fma(float(q), s, -z*s)) is the reference's value bit for bit.
#include "w4a16.cuh"

namespace mi355x {

// ------------------------------------------------------------------ repack / shuffle

// One thread per (k-block of 8 rows, word column): reads 8 AWQ words (rows 8kk..8kk+7 of
// word column nw) and writes the 8 exllama words (kk, 8nw .. 8nw+7).
__global__ void awq_to_gptq_4bit_kernel(uint32_t* __restrict__ out,
                                        const uint32_t* __restrict__ in, int k, int n) {
  const int n8 = n / 8;
  const int k8 = (k + 7) / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8 * k8) return;
  const int nw = idx % n8;   // consecutive threads -> consecutive input words (coalesced)
  const int kk = idx / n8;
  uint32_t a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = kk * 8 + j;
    a[j] = row < k ? in[(int64_t)row * n8 + nw] : 0u;
  }
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // output column 8nw + i
    uint32_t w = 0;
    const int sh = 4 * awq_shift(i);
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // k row 8kk + j goes to nibble awq_shift(j) (= exllama order)
      w |= ((a[j] >> sh) & 0xFu) << (4 * awq_shift(j));
    }
    o[i] = w;
  }
  uint32_t* dst = out + (int64_t)kk * n + nw * 8;
  *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[4], o[5], o[6], o[7]);
}

// exllama nibble shuffle of one word: nibbles (k0..k7) -> (k0,k2,k4,k6,k1,k3,k5,k7)
__device__ __forceinline__ uint32_t shuffle_4bit_word(uint32_t q) {
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r |= ((q >> (8 * i)) & 0xFu) << (4 * i);
    r |= ((q >> (8 * i + 4)) & 0xFu) << (4 * i + 16);
  }
  return r;
}

__global__ void gptq_shuffle_4bit_kernel(uint32_t* __restrict__ w, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) w[i] = shuffle_4bit_word(w[i]);
}

// 8-bit exllama shuffle (qdq_8.cuh: identity layout) — nothing to do per word.

// make_sequential: new row r of the UNPACKED matrix = old row q_perm[r].
__global__ void gptq_make_sequential_4bit_kernel(const uint32_t* __restrict__ w,
                                                 uint32_t* __restrict__ w_new,
                                                 const int* __restrict__ q_perm, int k8, int n) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row8 = blockIdx.y;
  if (col >= n) return;
  uint32_t dst = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int src_row = q_perm[row8 * 8 + i];
    const uint32_t src = w[(int64_t)(src_row >> 3) * n + col];
    dst |= ((src >> (4 * (src_row & 7))) & 0xFu) << (4 * i);
  }
  w_new[(int64_t)row8 * n + col] = dst;
}

__global__ void gptq_make_sequential_8bit_kernel(const uint32_t* __restrict__ w,
                                                 uint32_t* __restrict__ w_new,
                                                 const int* __restrict__ q_perm, int k4, int n) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row4 = blockIdx.y;
  if (col >= n) return;
  uint32_t dst = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int src_row = q_perm[row4 * 4 + i];
    const uint32_t src = w[(int64_t)(src_row >> 2) * n + col];
    dst |= ((src >> (8 * (src_row & 3))) & 0xFFu) << (8 * i);
  }
  w_new[(int64_t)row4 * n + col] = dst;
}

// ------------------------------------------------------------------- awq_dequantize
// Original AWQ layout: qweight [K][N/8], qzeros [K/g][N/8] (both N-interleaved), scales
// [K/g][N].  One thread per packed word -> 8 outputs (one 16-B store for 2-byte T).
template <typename T>
__global__ void awq_dequantize_kernel(T* __restrict__ out, const uint32_t* __restrict__ qw,
                                      const T* __restrict__ scales,
                                      const uint32_t* __restrict__ qz, int k, int n, int group) {
  const int n8 = n / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)k * n8) return;
  const int row = idx / n8;
  const int nw = idx - (int64_t)row * n8;
  const int g = row / group;
  const uint32_t w = qw[idx];
  const uint32_t z = qz[(int64_t)g * n8 + nw];
  T o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sh = 4 * awq_shift(j);
    const float q = (float)((w >> sh) & 0xFu);
    const float zz = (float)((z >> sh) & 0xFu);
    const float s = to_f32(scales[(int64_t)g * n + nw * 8 + j]);
    o[j] = from_f32<T>(fmaf(q, s, -zz * s));
  }
  T* dst = out + (int64_t)row * n + nw * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = o[j];
}

// ------------------------------------------------------------------- act-order gather
// out[m][kk] = in[m][perm[kk]]   (ref: perm_a, q_gemm.cu:1770-1786)
template <typename T>
__global__ void permute_cols_kernel(T* __restrict__ out, const T* __restrict__ in,
                                    const int* __restrict__ perm, int m, int k, int64_t lda) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (col < k) out[(int64_t)row * k + col] = in[(int64_t)row * lda + perm[col]];
}

// ------------------------------------------------------------------ small-M kernel
// C[M<=64, N] tile 64 x 64 per workgroup of 8 waves; the waves split the K range of the
// workgroup (k-step = 32, interleaved) and are summed through LDS (tree); blockIdx.y splits
// K across workgroups (fp32 partial slabs acc_ws[split][m][n] when gridDim.y > 1).
// Regime: weights are read exactly once (HBM-bound), the kernel is otherwise limited by the
// dequant VALU work and by memory latency, so it runs 8 waves x 2 workgroups per CU
// (<= 128 VGPRs) and lets wave-level parallelism hide the latency; weights are prefetched
// one k-step ahead (7 VGPRs), activations (<= 64 rows, L2 resident) are loaded per step.
constexpr int kSmBN = 64;
constexpr int kSmWaves = 8;
constexpr int kSmThreads = kSmWaves * 64;

template <typename T, int MT, int ZMODE, bool GPOW2, bool APACK>
__global__ __launch_bounds__(kSmThreads, 4) void w4a16_gemm_small_m_kernel(
    T* __restrict__ c, float* __restrict__ acc_ws, const T* __restrict__ a,
    const uint32_t* __restrict__ qw, const T* __restrict__ scales,
    const uint32_t* __restrict__ qz, int m, int n, int k, int group, int group_shift,
    int64_t lda, int ksteps_per_split, const uint4* __restrict__ a_packed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);  // [4][MT*16][64] fp32

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: uniform branches
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int n0 = blockIdx.x * kSmBN;
  const int ncol = n0 + 4 * lc;  // first of this lane's 4 columns

  const int total_steps = k / 32;
  const int step_begin = blockIdx.y * ksteps_per_split;
  const int step_end = min(step_begin + ksteps_per_split, total_steps);

  f32x4_t acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // Activation loads are COALESCED, not MFMA-shaped: lane l reads row 16*i + l/4, 16-byte
  // chunk l%4 of the k-step (4 consecutive lanes = 64 contiguous bytes).  The MFMA A operand
  // wants lane (lr, lc) = row lc, chunk lr, i.e. the value held by lane 4*lc + lr: a fixed lane
  // transpose done with ds_bpermute (LDS crossbar, no LDS memory).  Loading fragment-shaped
  // (16 different rows per 16 consecutive lanes) made the texture-address unit the bottleneck
  // (64 cache lines per wave-instruction instead of 8-16).
  const T* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int r = i * 16 + (lane >> 2);
    r = r < m ? r : m - 1;
    arow[i] = a + (int64_t)r * lda + 8 * (lane & 3);
  }
  const int perm_addr = (4 * lc + lr) * 4;  // ds_bpermute byte address of the source lane
  const int pk_slot = frag_swz(lr, lc);     // slot of this lane inside a packed operand image
  const int n8 = n >> 3;

  uint4 bq[2];
  uint2 scq[2];
  uint32_t zq[2];
  auto load_b = [&](int slot, int s) {
    const int kbase = s * 32;
    const int g = GPOW2 ? (kbase >> group_shift) : (kbase / group);
    bq[slot] = *reinterpret_cast<const uint4*>(qw + (int64_t)(4 * s + lr) * n + ncol);
    scq[slot] = *reinterpret_cast<const uint2*>(scales + (int64_t)g * n + ncol);
    zq[slot] = qz[(int64_t)g * n8 + (ncol >> 3)];
  };
  auto compute = [&](int slot, int s) {
    uint4 af[MT];
    if constexpr (APACK) {
      // activations pre-tiled into MFMA operand images: one lane-linear 1 KiB load per tile
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = a_packed[((int64_t)i * (k >> 5) + s) * 64 + pk_slot];
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const uint4 raw = *reinterpret_cast<const uint4*>(arow[i] + s * 32);
        af[i].x = __builtin_amdgcn_ds_bpermute(perm_addr, raw.x);
        af[i].y = __builtin_amdgcn_ds_bpermute(perm_addr, raw.y);
        af[i].z = __builtin_amdgcn_ds_bpermute(perm_addr, raw.z);
        af[i].w = __builtin_amdgcn_ds_bpermute(perm_addr, raw.w);
      }
    }
    T sct[4];
    *reinterpret_cast<uint2*>(sct) = scq[slot];
    float zp[4];
    unpack_zeros4<ZMODE>(zq[slot], ncol, zp);
    const uint32_t words[4] = {bq[slot].x, bq[slot].y, bq[slot].z, bq[slot].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float sc = to_f32(sct[t]);
      const uint4 bf = dequant_word<T>(words[t], sc, -zp[t] * sc);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][t] = Mfma<T>::run(af[i], bf, acc[i][t]);
    }
  };
  {
    // Every workgroup starts its K sweep at a different k-step (rotation `rot`): all
    // workgroups read the SAME activation rows, and sweeping them in lock-step concentrates
    // the requests on few L2 channels.  fp32 accumulation order changes, rounding points don't.
    const int cnt = step_end - step_begin;
    const int nsteps = wave < cnt ? (cnt - wave + kSmWaves - 1) / kSmWaves : 0;
    const int rot = (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)max(cnt, 1));
    auto step_of = [&](int i) {  // i-th k-step of this wave (clamped to its last one)
      const int ii = i < nsteps ? i : nsteps - 1;
      int idx = wave + kSmWaves * ii + rot;
      idx = idx >= cnt ? idx - cnt : idx;
      return step_begin + idx;
    };
    if (nsteps > 0) {
      load_b(0, step_of(0));
      for (int it = 0; it < nsteps; it += 2) {
        load_b(1, step_of(it + 1));
        __builtin_amdgcn_sched_barrier(0);  // keep the weight prefetch above the compute
        compute(0, step_of(it));
        if (it + 1 >= nsteps) break;
        load_b(0, step_of(it + 2));
        __builtin_amdgcn_sched_barrier(0);
        compute(1, step_of(it + 1));
      }
    }
  }

  // ---- sum the 8 waves: tree through LDS (4 slabs) ------------------------------------
  auto lds_store = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = i * 16 + 4 * lr + j;
        *reinterpret_cast<float4*>(dst + row * 64 + 4 * lc) =
            make_float4(acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]);
      }
    }
  };
  auto lds_add = [&](const float* src) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = i * 16 + 4 * lr + j;
        const float4 v = *reinterpret_cast<const float4*>(src + row * 64 + 4 * lc);
        acc[i][0][j] += v.x;
        acc[i][1][j] += v.y;
        acc[i][2][j] += v.z;
        acc[i][3][j] += v.w;
      }
    }
  };
  constexpr int kSlab = MT * 16 * 64;
#pragma unroll
  for (int half = kSmWaves / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) lds_store(red + (wave - half) * kSlab);
    __syncthreads();
    if (wave < half) lds_add(red + wave * kSlab);
    __syncthreads();
  }
  if (wave != 0) return;

  // ---- epilogue: lane holds, for each (i, j), 4 consecutive columns ncol..ncol+3 ----
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
        float* dst = acc_ws + ((int64_t)blockIdx.y * m + row) * n + ncol;
        *reinterpret_cast<float4*>(dst) =
            make_float4(acc[i][0][j], acc[i][1][j], acc[i][2][j], acc[i][3][j]);
      }
    }
  }
}

int w4a16_gemm_large_m_dispatch(const GemmArgs& g, int dtype);  // w4a16_large.hip
int w4a16_gemm_unfused_dispatch(const GemmArgs& g, int dtype);  // w4a16_unfused.hip
int w4a16_prepack_dispatch(const GemmArgs& g, int dtype);       // w4a16_unfused.hip
int w4a16_gemm_stripe_dispatch(const GemmArgs& g, int dtype, int row0, int rows);  // w4a16_stripe.hip
int w4a16_gemm_stripe_silu_dispatch(const GemmArgs& g, int dtype);                  // w4a16_stripe.hip

template <typename T, int ZMODE>
static int launch_small_m(const GemmArgs& g, int row0, int rows) {
  const T* a = static_cast<const T*>(g.a) + (int64_t)row0 * g.lda;
  T* c = static_cast<T*>(g.c) + (int64_t)row0 * g.n;
  const int col_tiles = g.n / kSmBN;
  const int total_steps = g.k / 32;
  // Split K across workgroups only when the column tiles alone cannot fill the chip: every
  // split writes a rows*n fp32 partial slab that the sum kernel reads back.  >= 16 k-steps per split.
  int sk = 1;
  if (g.ws != nullptr) {
    while (col_tiles * sk < 256 && total_steps / (sk * 2) >= 16 &&
           g.ws_elems >= (int64_t)(sk * 2) * rows * g.n)
      sk *= 2;
  }
  const int steps_per_split = (total_steps + sk - 1) / sk;
  sk = (total_steps + steps_per_split - 1) / steps_per_split;
  float* ws = g.ws;
  const int mt = (rows + 15) / 16;
  int group_shift = -1;  // log2(group) when group is a power of two (avoids an integer division)
  if ((g.group & (g.group - 1)) == 0) {
    group_shift = 0;
    while ((1 << group_shift) < g.group) ++group_shift;
  }
  // optional: re-tile the activations into MFMA operand images first (one tiny launch)
  const uint4* a_packed = nullptr;
  const int64_t pack_bytes = (int64_t)mt * 16 * g.k * 2;
  if (g.dq_ws != nullptr && g.dq_ws_bytes >= pack_bytes && g.k % 32 == 0) {
    hipLaunchKernelGGL(pack_a_kernel<T>, dim3((g.k + kPackK - 1) / kPackK, mt), dim3(256), 0, g.stream,
                       static_cast<T*>(g.dq_ws), a, rows, g.k, g.lda);
    int prc = check_launch("pack_a(small)");
    if (prc) return prc;
    a_packed = static_cast<const uint4*>(g.dq_ws);
  }
  dim3 grid(col_tiles, sk), block(kSmThreads);
#define LAUNCH_SM3(MTV, P2, AP)                                                               \
  hipLaunchKernelGGL((w4a16_gemm_small_m_kernel<T, MTV, ZMODE, P2, AP>), grid, block,         \
                     (size_t)(kSmWaves / 2) * MTV * 16 * 64 * sizeof(float), g.stream, c, ws, a, \
                     g.qw, static_cast<const T*>(g.scales), g.qz, rows, g.n, g.k, g.group,     \
                     group_shift, g.lda, steps_per_split, a_packed)
#define LAUNCH_SM2(MTV, P2)                    \
  do {                                         \
    if (a_packed) LAUNCH_SM3(MTV, P2, true);   \
    else LAUNCH_SM3(MTV, P2, false);           \
  } while (0)
#define LAUNCH_SM(MTV)                      \
  do {                                      \
    if (group_shift >= 0) LAUNCH_SM2(MTV, true); \
    else LAUNCH_SM2(MTV, false);            \
  } while (0)
  if (mt <= 1) LAUNCH_SM(1);
  else if (mt <= 2) LAUNCH_SM(2);
  else LAUNCH_SM(4);
#undef LAUNCH_SM3
#undef LAUNCH_SM2
#undef LAUNCH_SM
  int rc = check_launch("w4a16_gemm_small_m");
  if (rc) return rc;
  if (sk > 1) {
    const int64_t n4 = (int64_t)rows * g.n / 4;
    hipLaunchKernelGGL(w4a16_sum_slabs_kernel<T>, dim3((n4 + 255) / 256), dim3(256), 0, g.stream, c,
                       ws, n4, sk);
    rc = check_launch("w4a16_sum_slabs");
  }
  return rc;
}


// ------------------------------------------------------------------ any-N fallback
// Shapes the tiled kernels do not take (n % 64 != 0; the reference only asks n % 16 == 0,
// gemm_kernels.cu:194-201): one workgroup = 16 columns x 8 rows, 16 k-parts of one lane each,
// weights dequantised with the same exact formula, fp32 accumulation, LDS tree over the k-parts.
// Correctness path (tiny test shapes), not a performance kernel.
template <typename T, int ZMODE>
__global__ __launch_bounds__(256) void w4a16_gemm_any_n_kernel(
    T* __restrict__ c, const T* __restrict__ a, const uint32_t* __restrict__ qw,
    const T* __restrict__ scales, const uint32_t* __restrict__ qz, int m, int n, int k, int group,
    int64_t lda) {
  __shared__ float red[16][8][17];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int kp = threadIdx.x >> 4;           // k-part 0..15
  const int row0 = blockIdx.y * 8;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  if (col < n) {
    for (int kk = kp; kk < k / 8; kk += 16) {
      const int g = (kk * 8) / group;
      const uint32_t w = qw[(int64_t)kk * n + col];
      const float s = to_f32(scales[(int64_t)g * n + col]);
      float z;
      if constexpr (ZMODE == kZeroAwq) {
        z = (float)((qz[(int64_t)g * (n >> 3) + (col >> 3)] >> (4 * awq_shift(col & 7))) & 0xFu);
      } else {
        z = (float)(((qz[(int64_t)g * (n >> 3) + (col >> 3)] >> (4 * (col & 7))) & 0xFu) + 1u);
      }
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // k row 8kk + j sits at nibble awq_shift(j)
        const float q = (float)((w >> (4 * awq_shift(j))) & 0xFu);
        wv[j] = to_f32(from_f32<T>(fmaf(q, s, -z * s)));
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (row0 + r < m) {
          const T* ap = a + (int64_t)(row0 + r) * lda + kk * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r] = fmaf(to_f32(ap[j]), wv[j], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) red[kp][r][threadIdx.x & 15] = acc[r];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) sum += red[p][r][cc];
    const int oc = blockIdx.x * 16 + cc;
    if (row0 + r < m && oc < n) c[(int64_t)(row0 + r) * n + oc] = from_f32<T>(sum);
  }
}

// 8-bit GPTQ, any shape: same structure (16 columns x 8 rows per workgroup, 16 k-parts), 4 k per word.
// Decode-sized 8-bit GEMMs run here: a correctness path (the int4 stripe kernel has no 8-bit twin;
// m >= 1024 goes through the operand-image GEMM).
template <typename T>
__global__ __launch_bounds__(256) void w8a16_gemm_any_kernel(
    T* __restrict__ c, const T* __restrict__ a, const uint32_t* __restrict__ qw,
    const T* __restrict__ scales, const uint32_t* __restrict__ qz, int m, int n, int k, int group,
    int64_t lda) {
  __shared__ float red[16][8][17];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int kp = threadIdx.x >> 4;
  const int row0 = blockIdx.y * 8;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  if (col < n) {
    for (int kk = kp; kk < k / 4; kk += 16) {
      const int g = (kk * 4) / group;
      const uint32_t w = qw[(int64_t)kk * n + col];
      const float s = to_f32(scales[(int64_t)g * n + col]);
      const float z = (float)(((qz[(int64_t)g * (n >> 2) + (col >> 2)] >> (8 * (col & 3))) & 0xFFu) + 1u);
      float wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = to_f32(from_f32<T>(fmaf((float)((w >> (8 * j)) & 0xFFu), s, -z * s)));
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (row0 + r < m) {
          const T* ap = a + (int64_t)(row0 + r) * lda + kk * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r] = fmaf(to_f32(ap[j]), wv[j], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) red[kp][r][threadIdx.x & 15] = acc[r];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int r = threadIdx.x >> 4, cc = threadIdx.x & 15;
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) sum += red[p][r][cc];
    const int oc = blockIdx.x * 16 + cc;
    if (row0 + r < m && oc < n) c[(int64_t)(row0 + r) * n + oc] = from_f32<T>(sum);
  }
}

template <typename T>
static int launch_any_n(const GemmArgs& g) {
  if (g.bits == 8) {
    dim3 grid8((g.n + 15) / 16, (g.m + 7) / 8), block8(256);
    hipLaunchKernelGGL(w8a16_gemm_any_kernel<T>, grid8, block8, 0, g.stream, static_cast<T*>(g.c),
                       static_cast<const T*>(g.a), g.qw, static_cast<const T*>(g.scales), g.qz, g.m, g.n,
                       g.k, g.group, g.lda);
    return check_launch("w8a16_gemm_any");
  }
  dim3 grid((g.n + 15) / 16, (g.m + 7) / 8), block(256);
  if (g.zmode == kZeroAwq) {
    hipLaunchKernelGGL((w4a16_gemm_any_n_kernel<T, kZeroAwq>), grid, block, 0, g.stream,
                       static_cast<T*>(g.c), static_cast<const T*>(g.a), g.qw,
                       static_cast<const T*>(g.scales), g.qz, g.m, g.n, g.k, g.group, g.lda);
  } else {
    hipLaunchKernelGGL((w4a16_gemm_any_n_kernel<T, kZeroGptq>), grid, block, 0, g.stream,
                       static_cast<T*>(g.c), static_cast<const T*>(g.a), g.qw,
                       static_cast<const T*>(g.scales), g.qz, g.m, g.n, g.k, g.group, g.lda);
  }
  return check_launch("w4a16_gemm_any_n");
}

template <typename T>
static int run_gemm_t(const GemmArgs& g, int dtype) {
  if (g.n % 64 != 0) {
    MI355X_REQUIRE(g.defer_sk == nullptr && !g.fuse_silu && !g.a_packed, MI355X_EUNSUPPORTED,
                   "w4a16 gemm: the fused / deferred forms need n %% 64 == 0 (n = %d)", g.n);
    return launch_any_n<T>(g);
  }
  if (g.bits == 8) {
    if (g.m >= 1024) {
      int rc = w4a16_gemm_unfused_dispatch(g, dtype);
      if (rc != 1) return rc;
    }
    return launch_any_n<T>(g);
  }
  if (g.m >= 1024) {
    int rc = w4a16_gemm_unfused_dispatch(g, dtype);
    if (rc != 1) return rc;  // 1 = no scratch / shape not handled: fall through to the fused kernel
  }
  // 64 < m < 1024 (chunked-prefill budgets of a few hundred tokens, mixed batches): passes of up to 128 rows
  // through the stripe kernel, split-K slabs in the fp32 workspace or, when the caller gave none, in the scratch
  // of the prefill path — measured (scripts/bench_gemm.py, Llama-3-8B layer): 1040-1130 us per layer with the
  // 128 x 256-tile kernel below (16-48 workgroups for the narrow projections) against ~90 us per 128-row pass.
  GemmArgs gm = g;
  if (g.m > 64 && gm.ws == nullptr && g.dq_ws != nullptr && (reinterpret_cast<uintptr_t>(g.dq_ws) & 15) == 0) {
    gm.ws = static_cast<float*>(g.dq_ws);
    gm.ws_elems = g.dq_ws_bytes / 4;
  }
  const int pass = g.m > 64 ? 128 : 64;
  bool stripe_ok = true;
  if (g.m >= 128) {
    // probe: does the stripe path take this shape at all?  (otherwise the tile kernel below serves it)
    stripe_ok = g.k % 128 == 0 && g.n % 64 == 0 && (g.group % 128 == 0 || (g.group >= 32 && 128 % g.group == 0));
    if (!stripe_ok) {
      int rc = w4a16_gemm_large_m_dispatch(g, dtype);
      if (rc != 1) return rc;  // 1 = shape not handled by the large-M kernel
    }
  }
  for (int row0 = 0; row0 < g.m; row0 += pass) {
    const int rows = (g.m - row0) < pass ? (g.m - row0) : pass;
    int rc = w4a16_gemm_stripe_dispatch(gm, dtype, row0, rows);
    if (rc == 1 && rows > 64) {       // not a stripe shape after all: two 64-row halves on the general kernels
      for (int r1 = row0; r1 < row0 + rows; r1 += 64) {
        const int rr = (row0 + rows - r1) < 64 ? (row0 + rows - r1) : 64;
        rc = w4a16_gemm_stripe_dispatch(gm, dtype, r1, rr);
        if (rc == 1)
          rc = g.zmode == kZeroAwq ? launch_small_m<T, kZeroAwq>(gm, r1, rr) : launch_small_m<T, kZeroGptq>(gm, r1, rr);
        if (rc) return rc;
      }
      continue;
    }
    if (rc != 1) {
      if (rc) return rc;
      continue;
    }
    rc = g.zmode == kZeroAwq ? launch_small_m<T, kZeroAwq>(g, row0, rows)
                                 : launch_small_m<T, kZeroGptq>(g, row0, rows);
    if (rc) return rc;
  }
  return MI355X_OK;
}

static int validate_gemm(const GemmArgs& g, const char* name) {
  MI355X_REQUIRE(g.m >= 0 && g.n > 0 && g.k > 0 && g.group > 0, MI355X_EINVAL, "%s: bad sizes", name);
  // the reference asks n % 16 == 0, k % 32 == 0 (gemm_kernels.cu:194-201); n % 64 != 0 runs a fallback
  MI355X_REQUIRE(g.n % 8 == 0, MI355X_EUNSUPPORTED, "%s: n = %d must be a multiple of 8", name, g.n);
  MI355X_REQUIRE(g.k % 32 == 0, MI355X_EUNSUPPORTED, "%s: k = %d must be a multiple of 32", name, g.k);
  MI355X_REQUIRE(g.group % 32 == 0 && g.k % g.group == 0, MI355X_EUNSUPPORTED,
                 "%s: group_size %d must be a multiple of 32 that divides k", name, g.group);
  MI355X_REQUIRE(g.lda % 8 == 0 && (reinterpret_cast<uintptr_t>(g.a) & 15) == 0, MI355X_EUNSUPPORTED,
                 "%s: activations must be 16-byte aligned with lda %% 8 == 0", name);
  if (g.m == 0) return MI355X_OK;
  MI355X_REQUIRE(g.c && g.a && g.qw && g.scales && g.qz, MI355X_EINVAL, "%s: null pointer", name);
  return MI355X_OK;
}

}  // namespace mi355x

using namespace mi355x;

extern "C" {

int mi355x_awq_to_gptq_4bit(uint32_t* out, const uint32_t* qweight, int k, int n,
                            mi355x_stream stream) {
  MI355X_REQUIRE(k > 0 && n > 0 && n % 8 == 0, MI355X_EINVAL,
                 "awq_to_gptq_4bit: bad sizes k=%d n=%d", k, n);
  MI355X_REQUIRE(out && qweight, MI355X_EINVAL, "awq_to_gptq_4bit: null pointer");
  MI355X_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, MI355X_EINVAL,
                 "awq_to_gptq_4bit: out must be 16-byte aligned");
  const int total = (n / 8) * ((k + 7) / 8);
  hipLaunchKernelGGL(awq_to_gptq_4bit_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), out, qweight, k, n);
  return check_launch("awq_to_gptq_4bit");
}

int mi355x_awq_dequantize(void* out, const uint32_t* qweight, const void* scales,
                          const uint32_t* qzeros, int k, int n, int group_size,
                          int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(k > 0 && n > 0 && n % 8 == 0 && group_size > 0, MI355X_EINVAL,
                 "awq_dequantize: bad sizes");
  MI355X_REQUIRE(out && qweight && scales && qzeros, MI355X_EINVAL, "awq_dequantize: null pointer");
  return MI355X_DISPATCH_HALF(dtype, [&] {
    const int64_t total = (int64_t)k * (n / 8);
    hipLaunchKernelGGL(awq_dequantize_kernel<scalar_t>, dim3((total + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<scalar_t*>(out), qweight,
                       static_cast<const scalar_t*>(scales), qzeros, k, n, group_size);
    return check_launch("awq_dequantize");
  });
}

int mi355x_gptq_shuffle(uint32_t* q_weight, const int* q_perm, uint32_t* scratch, int k,
                        int n, int bit, mi355x_stream stream) {
  MI355X_REQUIRE(bit == 4 || bit == 8, MI355X_EUNSUPPORTED,
                 "gptq_shuffle: only 4- and 8-bit weights are supported (bit=%d)", bit);
  MI355X_REQUIRE(k > 0 && n > 0 && k % 32 == 0, MI355X_EINVAL, "gptq_shuffle: bad sizes");
  MI355X_REQUIRE(q_weight, MI355X_EINVAL, "gptq_shuffle: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rows = k / 32 * bit;  // packed rows
  if (q_perm) {
    MI355X_REQUIRE(scratch, MI355X_EINVAL, "gptq_shuffle: q_perm given but no scratch buffer");
    dim3 grid((n + 255) / 256, rows), block(256);
    if (bit == 4)
      hipLaunchKernelGGL(gptq_make_sequential_4bit_kernel, grid, block, 0, s, q_weight, scratch,
                         q_perm, rows, n);
    else
      hipLaunchKernelGGL(gptq_make_sequential_8bit_kernel, grid, block, 0, s, q_weight, scratch,
                         q_perm, rows, n);
    int rc = check_launch("gptq_make_sequential");
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(q_weight, scratch, (size_t)rows * n * sizeof(uint32_t),
                                  hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) {
      set_error("gptq_shuffle: hipMemcpyAsync: %s", hipGetErrorString(e));
      return MI355X_ELAUNCH;
    }
  }
  if (bit == 4) {
    const int64_t total = (int64_t)rows * n;
    hipLaunchKernelGGL(gptq_shuffle_4bit_kernel, dim3((total + 255) / 256), dim3(256), 0, s,
                       q_weight, total);
    return check_launch("gptq_shuffle");
  }
  return MI355X_OK;
}

int mi355x_awq_gemm(void* c, const void* a, const uint32_t* qweight, const void* scales,
                    const uint32_t* qzeros, float* workspace, int64_t workspace_elems,
                    void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k,
                    int group_size, int64_t lda, int dtype, mi355x_stream stream) {
  GemmArgs g{c, a, qweight, scales, qzeros, workspace, workspace_elems, dq_workspace,
             dq_workspace_bytes, m, n, k, group_size, lda, kZeroAwq,
             static_cast<hipStream_t>(stream)};
  int rc = validate_gemm(g, "awq_gemm");
  if (rc || m == 0) return rc;
  return MI355X_DISPATCH_HALF(dtype, [&] { return run_gemm_t<scalar_t>(g, dtype); });
}

int mi355x_awq_gemm_deferred(void* c, const void* a, const uint32_t* qweight, const void* scales,
                             const uint32_t* qzeros, float* workspace, int64_t workspace_elems,
                             void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k,
                             int group_size, int64_t lda, int dtype, int* sk_out,
                             mi355x_stream stream) {
  MI355X_REQUIRE(sk_out != nullptr, MI355X_EINVAL, "awq_gemm_deferred: sk_out is null");
  *sk_out = 0;
  GemmArgs g{c, a, qweight, scales, qzeros, workspace, workspace_elems, dq_workspace,
             dq_workspace_bytes, m, n, k, group_size, lda, kZeroAwq,
             static_cast<hipStream_t>(stream)};
  g.defer_sk = sk_out;
  int rc = validate_gemm(g, "awq_gemm_deferred");
  if (rc || m == 0) return rc;
  return MI355X_DISPATCH_HALF(dtype, [&] { return run_gemm_t<scalar_t>(g, dtype); });
}

int mi355x_awq_gemm_silu_mul(void* out, const void* a, const uint32_t* qweight, const void* scales,
                             const uint32_t* qzeros, void* dq_workspace, int64_t dq_workspace_bytes,
                             int m, int n, int k, int group_size, int64_t lda, int dtype,
                             mi355x_stream stream) {
  GemmArgs g{out, a, qweight, scales, qzeros, nullptr, 0, dq_workspace, dq_workspace_bytes,
             m, n, k, group_size, lda, kZeroAwq, static_cast<hipStream_t>(stream)};
  g.fuse_silu = true;
  int rc = validate_gemm(g, "awq_gemm_silu_mul");
  if (rc || m == 0) return rc;
  rc = m <= 64 ? w4a16_gemm_stripe_silu_dispatch(g, dtype) : w4a16_gemm_unfused_dispatch(g, dtype);
  MI355X_REQUIRE(rc != 1, MI355X_EUNSUPPORTED,
                 "awq_gemm_silu_mul: needs a 2-byte dtype and either m <= 64 (n %% 128 == 0, k %% 128 == 0, "
                 "group 32/64/128k) or m >= 1024 (n %% 256 == 0, k %% 32 == 0, dq_workspace >= "
                 "(n + roundup(m,16))*k*2 bytes); got m=%d n=%d k=%d, %lld bytes; use awq_gemm + silu_and_mul",
                 m, n, k, (long long)dq_workspace_bytes);
  return rc;
}

int mi355x_awq_gemm_silu_mul_packed(void* out_packed, const void* a, const uint32_t* qweight,
                                    const void* scales, const uint32_t* qzeros, void* dq_workspace,
                                    int64_t dq_workspace_bytes, int m, int n, int k, int group_size,
                                    int64_t lda, int dtype, mi355x_stream stream) {
  GemmArgs g{out_packed, a, qweight, scales, qzeros, nullptr, 0, dq_workspace, dq_workspace_bytes,
             m, n, k, group_size, lda, kZeroAwq, static_cast<hipStream_t>(stream)};
  g.fuse_silu = true;
  g.out_packed = true;
  int rc = validate_gemm(g, "awq_gemm_silu_mul_packed");
  if (rc || m == 0) return rc;
  MI355X_REQUIRE((reinterpret_cast<uintptr_t>(out_packed) & 15) == 0, MI355X_EUNSUPPORTED,
                 "awq_gemm_silu_mul_packed: out_packed must be 16-byte aligned");
  rc = w4a16_gemm_unfused_dispatch(g, dtype);
  MI355X_REQUIRE(rc != 1, MI355X_EUNSUPPORTED,
                 "awq_gemm_silu_mul_packed: needs a 2-byte dtype, m >= 1024, n %% 256 == 0, k %% 32 == 0 "
                 "and dq_workspace >= (n + roundup(m,16))*k*2 bytes; got m=%d n=%d k=%d, %lld bytes",
                 m, n, k, (long long)dq_workspace_bytes);
  return rc;
}

int mi355x_awq_gemm_packed_a(void* out, const void* a_packed, const uint32_t* qweight,
                             const void* scales, const uint32_t* qzeros, void* dq_workspace,
                             int64_t dq_workspace_bytes, int m, int n, int k, int group_size,
                             int dtype, mi355x_stream stream) {
  GemmArgs g{out, a_packed, qweight, scales, qzeros, nullptr, 0, dq_workspace, dq_workspace_bytes,
             m, n, k, group_size, k, kZeroAwq, static_cast<hipStream_t>(stream)};
  g.a_packed = true;
  int rc = validate_gemm(g, "awq_gemm_packed_a");
  if (rc || m == 0) return rc;
  rc = w4a16_gemm_unfused_dispatch(g, dtype);
  MI355X_REQUIRE(rc != 1, MI355X_EUNSUPPORTED,
                 "awq_gemm_packed_a: needs a 2-byte dtype, m >= 1024, n %% 64 == 0, k %% 32 == 0 and "
                 "dq_workspace >= (n + roundup(m,16))*k*2 bytes; got m=%d n=%d k=%d, %lld bytes",
                 m, n, k, (long long)dq_workspace_bytes);
  return rc;
}

int mi355x_w4a16_prepack(void* image, const uint32_t* qweight, const void* scales,
                         const uint32_t* qzeros, int n, int k, int group_size, int gptq_zeros, int dtype,
                         mi355x_stream stream) {
  MI355X_REQUIRE(n > 0 && k > 0 && group_size > 0 && n % 64 == 0 && k % 32 == 0 &&
                     group_size % 32 == 0 && k % group_size == 0,
                 MI355X_EUNSUPPORTED, "w4a16_prepack: needs n %% 64 == 0, k %% 32 == 0, group %% 32 == 0 "
                 "(n=%d k=%d group=%d)", n, k, group_size);
  MI355X_REQUIRE(image && qweight && scales && qzeros, MI355X_EINVAL, "w4a16_prepack: null pointer");
  MI355X_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, MI355X_EINVAL,
                 "w4a16_prepack: image must be 16-byte aligned");
  GemmArgs g{image, nullptr, qweight, scales, qzeros, nullptr, 0, nullptr, 0, 0, n, k, group_size, 0,
             gptq_zeros ? kZeroGptq : kZeroAwq, static_cast<hipStream_t>(stream)};
  const int rc = w4a16_prepack_dispatch(g, dtype);
  MI355X_REQUIRE(rc != 1, MI355X_EUNSUPPORTED, "w4a16_prepack: needs a 2-byte dtype");
  return rc;
}

int mi355x_w4a16_gemm_prepacked(void* out, const void* a, const void* image, void* a_workspace,
                                int64_t a_workspace_bytes, int m, int n, int k, int64_t lda, int mode,
                                int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(m >= 0 && n > 0 && k > 0 && n % 64 == 0 && k % 32 == 0, MI355X_EUNSUPPORTED,
                 "w4a16_gemm_prepacked: bad sizes m=%d n=%d k=%d", m, n, k);
  if (m == 0) return MI355X_OK;
  MI355X_REQUIRE(out && a && image, MI355X_EINVAL, "w4a16_gemm_prepacked: null pointer");
  const bool silu = mode & MI355X_PREPACKED_SILU, outp = mode & MI355X_PREPACKED_OUT_IMAGE,
             ain = mode & MI355X_PREPACKED_A_IMAGE;
  MI355X_REQUIRE(!outp || silu, MI355X_EINVAL, "w4a16_gemm_prepacked: OUT_IMAGE needs SILU");
  GemmArgs g{out, a, nullptr, nullptr, nullptr, nullptr, 0, a_workspace, a_workspace_bytes, m, n, k,
             32, ain ? (int64_t)k : lda, kZeroAwq, static_cast<hipStream_t>(stream)};
  g.fuse_silu = silu;
  g.out_packed = outp;
  g.a_packed = ain;
  g.b_image = image;
  if (!ain) {
    MI355X_REQUIRE(lda % 8 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0, MI355X_EUNSUPPORTED,
                   "w4a16_gemm_prepacked: activations must be 16-byte aligned with lda %% 8 == 0");
  }
  const int rc = w4a16_gemm_unfused_dispatch(g, dtype);
  MI355X_REQUIRE(rc != 1, MI355X_EUNSUPPORTED,
                 "w4a16_gemm_prepacked: needs a 2-byte dtype, m >= 384 (n %% 256 == 0 with SILU) and an "
                 "activation workspace of roundup(m,16)*k*2 bytes (got m=%d n=%d k=%d, %lld bytes)",
                 m, n, k, (long long)a_workspace_bytes);
  return rc;
}

// 4-byte workspace elements the operand-image GEMM wants ON TOP of its operand images to split K for a shape with
// few 256 x 256 tiles (0: no split)
int64_t mi355x_w4a16_prepacked_split_elems(int m, int n, int k) {
  if (m < kW4PrepackedMinM || n <= 0 || k <= 0 || n % 64 != 0 || k % 32 != 0 || n % 16 != 0) return 0;
  const int sk = plan_tile_split(m, n, k / 64, INT64_MAX, 0);
  return sk > 1 ? (int64_t)sk * m * n : 0;
}

int mi355x_gptq_gemm(void* c, const void* a, const uint32_t* qweight,
                     const uint32_t* qzeros, const void* scales, const int* g_idx,
                     void* perm_space, float* workspace, int64_t workspace_elems,
                     void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k, int bit,
                     int group_size, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(bit == 4 || bit == 8, MI355X_EUNSUPPORTED,
                 "gptq_gemm: 4-bit and 8-bit weights are implemented (bit=%d)", bit);
  GemmArgs g{c, a, qweight, scales, qzeros, workspace, workspace_elems, dq_workspace,
             dq_workspace_bytes, m, n, k, group_size, k, kZeroGptq,
             static_cast<hipStream_t>(stream)};
  g.bits = bit;
  int rc = validate_gemm(g, "gptq_gemm");
  if (rc || m == 0) return rc;
  if (g_idx) {
    MI355X_REQUIRE(perm_space, MI355X_EINVAL, "gptq_gemm: act-order needs perm_space");
    rc = MI355X_DISPATCH_HALF(dtype, [&] {
      hipLaunchKernelGGL(permute_cols_kernel<scalar_t>, dim3((k + 255) / 256, m), dim3(256), 0,
                         g.stream, static_cast<scalar_t*>(perm_space),
                         static_cast<const scalar_t*>(a), g_idx, m, k, (int64_t)k);
      return check_launch("gptq_gemm_permute");
    });
    if (rc) return rc;
    g.a = perm_space;
  }
  return MI355X_DISPATCH_HALF(dtype, [&] { return run_gemm_t<scalar_t>(g, dtype); });
}

}  // extern "C"
