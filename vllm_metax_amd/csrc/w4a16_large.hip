// w4a16_large.hip — w4a16 dequant-GEMM for prefill-sized M (>= 128 rows) on gfx950 MFMA.
//
// Same arithmetic as the small-M kernel (w4a16.hip / w4a16.cuh; reference citations there):
// weights are dequantised exactly to scalar_t, products accumulate in fp32 on
// v_mfma_f32_16x16x32_{bf16,f16}, C is rounded to scalar_t once.
//
// Tile: 128 (M) x 256 (N) per 256-thread workgroup, K step 64.
//   * each of the 4 waves owns 64 columns and ALL 128 rows, so every int4 word is
//     dequantised exactly once per workgroup; its B operand comes straight from global
//     memory into VGPRs (one 16-B load per lane and 32-deep k-step, see w4a16.hip);
//   * the A tile (shared by the 4 waves) is read from global memory in full 128-byte lines
//     (8 lanes per row) and staged through LDS in FRAGMENT-MAJOR order: piece (mt, ks) is
//     the 1 KiB image of one 16x32 MFMA A operand, so a fragment is one ds_read_b128 per
//     lane (issue-early / write-late double buffering);
//   * blockIdx is remapped so that the workgroups resident on one XCD share a B panel.
#include "w4a16.cuh"

namespace mi355x {

constexpr int kLgBM = 128;
constexpr int kLgBN = 256;
constexpr int kLgBK = 64;
constexpr int kLgThreads = 256;
constexpr int kLgMT = kLgBM / 16;  // 8 M-tiles per wave

template <typename T, int ZMODE>
__global__ __launch_bounds__(kLgThreads, 2) void w4a16_gemm_large_m_kernel(
    T* __restrict__ c, const T* __restrict__ a, const uint32_t* __restrict__ qw,
    const T* __restrict__ scales, const uint32_t* __restrict__ qz, int m, int n, int k,
    int group, int64_t lda, int num_m_blocks, int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* a_lds = reinterpret_cast<uint4*>(smem);  // [2][16 pieces][64 lanes]

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give each XCD a contiguous
  // run of tiles (tiles are ordered m-fastest, so a run shares its B panel).
  int tile;
  {
    const int b = blockIdx.x;
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

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lc = lane & 15;
  const int lr = lane >> 4;
  const int ncol = nb * kLgBN + wave * 64 + 4 * lc;
  const bool wave_active = (nb * kLgBN + wave * 64) < n;
  const int n8 = n >> 3;
  const int ktiles = k / kLgBK;

  // A staging: COALESCED global loads — thread t reads row t/8 (+32 per pass), 16-byte chunk
  // t%8 of the row's 128-byte K-tile segment (8 consecutive lanes = one full 128-B line) — and
  // writes it to the fragment-major LDS image: piece (mt = row/16, ks = chunk/4), slot
  // (row%16)*4 + chunk%4, so that the MFMA A fragment of lane (lr, lc) is slot lc*4 + lr.
  const int st_row = threadIdx.x >> 3;   // 0..31
  const int st_chunk = threadIdx.x & 7;  // 0..7
  const T* a_src[4];
  int a_dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lrow = st_row + 32 * i;    // row inside the 128-row tile
    int row = mb * kLgBM + lrow;
    row = row < m ? row : m - 1;
    a_src[i] = a + (int64_t)row * lda + 8 * st_chunk;
    a_dst[i] = ((lrow >> 4) * 2 + (st_chunk >> 2)) * 64 + (lrow & 15) * 4 + (st_chunk & 3);
  }
  const int frag_slot = lc * 4 + lr;

  f32x4_t acc[kLgMT][4];
#pragma unroll
  for (int i = 0; i < kLgMT; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  uint4 a_stage[4];
  uint4 b_cur[2], b_nxt[2];
  auto load_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_stage[i] = *reinterpret_cast<const uint4*>(a_src[i] + kt * kLgBK);
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_lds[buf * 1024 + a_dst[i]] = a_stage[i];
  };
  auto load_b = [&](int kt, uint4 (&dst)[2]) {
    if (wave_active) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dst[ks] = *reinterpret_cast<const uint4*>(qw + (int64_t)(8 * kt + 4 * ks + lr) * n + ncol);
      }
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
      for (int ks = 0; ks < 2; ++ks) {
        const int g = (kt * kLgBK + ks * 32) / group;
        float sc[4], zp[4];
        {
          const T* sp = scales + (int64_t)g * n + ncol;
#pragma unroll
          for (int t = 0; t < 4; ++t) sc[t] = to_f32(sp[t]);
          load_zeros4<ZMODE>(qz + (int64_t)g * n8, ncol, zp);
        }
        const uint32_t words[4] = {b_cur[ks].x, b_cur[ks].y, b_cur[ks].z, b_cur[ks].w};
        uint4 bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = dequant_word<T>(words[t], sc[t], -zp[t] * sc[t]);
#pragma unroll
        for (int i = 0; i < kLgMT; ++i) {
          const uint4 af = a_lds[(cur * 16 + i * 2 + ks) * 64 + frag_slot];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[i][t] = Mfma<T>::run(af, bf[t], acc[i][t]);
        }
      }
    }
    if (has_next) {
      store_a(cur ^ 1);
      b_cur[0] = b_nxt[0];
      b_cur[1] = b_nxt[1];
    }
    __syncthreads();
    cur ^= 1;
  }

  if (!wave_active) return;
#pragma unroll
  for (int i = 0; i < kLgMT; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = mb * kLgBM + i * 16 + 4 * lr + j;
      if (row < m) {
        const uint2 v = make_uint2(Mfma<T>::pack(acc[i][0][j], acc[i][1][j]),
                                   Mfma<T>::pack(acc[i][2][j], acc[i][3][j]));
        *reinterpret_cast<uint2*>(c + (int64_t)row * n + ncol) = v;
      }
    }
  }
}

template <typename T>
static int launch_large(const GemmArgs& g) {
  const int num_m_blocks = (g.m + kLgBM - 1) / kLgBM;
  const int num_n_blocks = (g.n + kLgBN - 1) / kLgBN;
  const int num_tiles = num_m_blocks * num_n_blocks;
  const size_t smem = 2 * 16 * 64 * sizeof(uint4);  // 32 KiB
  dim3 grid(num_tiles), block(kLgThreads);
  if (g.zmode == kZeroAwq) {
    hipLaunchKernelGGL((w4a16_gemm_large_m_kernel<T, kZeroAwq>), grid, block, smem, g.stream,
                       static_cast<T*>(g.c), static_cast<const T*>(g.a), g.qw,
                       static_cast<const T*>(g.scales), g.qz, g.m, g.n, g.k, g.group, g.lda,
                       num_m_blocks, num_tiles);
  } else {
    hipLaunchKernelGGL((w4a16_gemm_large_m_kernel<T, kZeroGptq>), grid, block, smem, g.stream,
                       static_cast<T*>(g.c), static_cast<const T*>(g.a), g.qw,
                       static_cast<const T*>(g.scales), g.qz, g.m, g.n, g.k, g.group, g.lda,
                       num_m_blocks, num_tiles);
  }
  return check_launch("w4a16_gemm_large_m");
}

// returns 1 when the shape is not handled here (caller falls back to the small-M kernel)
int w4a16_gemm_large_m_dispatch(const GemmArgs& g, int dtype) {
  if (g.k % kLgBK != 0) return 1;
  if (dtype == MI355X_BF16) return launch_large<bf16_t>(g);
  if (dtype == MI355X_F16) return launch_large<f16_t>(g);
  return 1;
}

}  // namespace mi355x
