/*
 * cpu_port.c — plain-C (OpenMP) restatement of the hot-path ops on the Llama decode step.
 * TEST INFRASTRUCTURE ONLY: used (a) by tests/ as a second, independently written oracle
 * that is cross-checked against oracle/ref_ops.py and the golden vectors, and (b) by
 * bench.py's `cpu_baseline` leg (kind "port"), timed on the GPU box's host cores.  The
 * product package never links or loads it.
 *
 * bf16 only (the bench dtype).  Each function cites the reference source it follows
 * (paths relative to /root/reference); rounding points are the reference's:
 * every store to a bf16 value is round-to-nearest-even of an fp32 intermediate.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -march=native -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint16_t bf16;

static inline float bf2f(bf16 v) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)v << 16;
  return c.f;
}
/* round-to-nearest-even, NaN preserved (c10::BFloat16 round_to_nearest_even semantics) */
static inline bf16 f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (bf16)((c.u >> 16) | 0x0040u);
  uint32_t lsb = (c.u >> 16) & 1u;
  c.u += 0x7fffu + lsb;
  return (bf16)(c.u >> 16);
}

/* cap the OpenMP team (a GPU box shows every hardware thread of the host but grants a share) */
void cpu_port_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int cpu_port_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp single
    n = omp_get_num_threads();
  }
#endif
  return n;
}

/* ---- rms_norm / fused_add_rms_norm: csrc/layernorm_kernels.cu:12-41, 104-137 ---------- */
void cpu_rms_norm_bf16(bf16* out, const bf16* in, const bf16* w, float eps, int tokens, int hidden) {
#pragma omp parallel for
  for (int t = 0; t < tokens; ++t) {
    const bf16* x = in + (size_t)t * hidden;
    float ss = 0.f;
    for (int i = 0; i < hidden; ++i) { float v = bf2f(x[i]); ss += v * v; }
    const float inv = 1.0f / sqrtf(ss / hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      const bf16 n = f2bf(bf2f(x[i]) * inv);
      out[(size_t)t * hidden + i] = f2bf(bf2f(n) * bf2f(w[i]));
    }
  }
}

void cpu_fused_add_rms_norm_bf16(bf16* in, bf16* res, const bf16* w, float eps, int tokens, int hidden) {
#pragma omp parallel for
  for (int t = 0; t < tokens; ++t) {
    bf16* x = in + (size_t)t * hidden;
    bf16* r = res + (size_t)t * hidden;
    float ss = 0.f;
    for (int i = 0; i < hidden; ++i) {
      const bf16 z = f2bf(bf2f(x[i]) + bf2f(r[i]));
      r[i] = z;
      const float v = bf2f(z);
      ss += v * v;
    }
    const float inv = 1.0f / sqrtf(ss / hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      const bf16 n = f2bf(bf2f(r[i]) * inv);
      x[i] = f2bf(bf2f(n) * bf2f(w[i]));
    }
  }
}

/* ---- rotary (NeoX): csrc/pos_encoding_kernels.cu:10-34, 37-100 -------------------------- */
void cpu_rotary_neox_bf16(const int64_t* pos, bf16* q, bf16* k, const bf16* cache, int tokens,
                          int rot_dim, int64_t q_stride, int64_t k_stride, int heads, int kv_heads,
                          int head_size) {
  const int emb = rot_dim / 2;
#pragma omp parallel for
  for (int t = 0; t < tokens; ++t) {
    const bf16* cs = cache + pos[t] * rot_dim;
    for (int h = 0; h < heads + kv_heads; ++h) {
      bf16* base = h < heads ? q + t * q_stride + (int64_t)h * head_size
                             : k + t * k_stride + (int64_t)(h - heads) * head_size;
      for (int i = 0; i < emb; ++i) {
        const bf16 x = base[i], y = base[emb + i], c = cs[i], s = cs[emb + i];
        const bf16 xc = f2bf(bf2f(x) * bf2f(c)), ys = f2bf(bf2f(y) * bf2f(s));
        const bf16 yc = f2bf(bf2f(y) * bf2f(c)), xs = f2bf(bf2f(x) * bf2f(s));
        base[i] = f2bf(bf2f(xc) - bf2f(ys));
        base[emb + i] = f2bf(bf2f(yc) + bf2f(xs));
      }
    }
  }
}

/* ---- silu_and_mul: csrc/activation_kernels.cu:14-36, 142-147 ---------------------------- */
void cpu_silu_and_mul_bf16(bf16* out, const bf16* in, int tokens, int d) {
#pragma omp parallel for
  for (int t = 0; t < tokens; ++t) {
    for (int i = 0; i < d; ++i) {
      const float x = bf2f(in[(size_t)t * 2 * d + i]);
      const bf16 s = f2bf(x / (1.0f + expf(-x)));
      out[(size_t)t * d + i] = f2bf(bf2f(s) * bf2f(in[(size_t)t * 2 * d + d + i]));
    }
  }
}

/* ---- reshape_and_cache (x-split layout): csrc/cache_kernels.cu:203-255 ------------------ */
void cpu_reshape_and_cache_bf16(const bf16* key, const bf16* value, bf16* kc, bf16* vc,
                                const int64_t* slots, int tokens, int64_t k_stride, int64_t v_stride,
                                int heads, int head_size, int block_size) {
  const int x = 8;
  for (int t = 0; t < tokens; ++t) {
    const int64_t slot = slots[t];
    if (slot < 0) continue;
    const int64_t blk = slot / block_size, off = slot % block_size;
    for (int i = 0; i < heads * head_size; ++i) {
      const int h = i / head_size, ho = i % head_size;
      kc[(((blk * heads + h) * (head_size / x) + ho / x) * block_size + off) * x + ho % x] =
          key[t * k_stride + i];
      vc[((blk * heads + h) * head_size + ho) * block_size + off] = value[t * v_stride + i];
    }
  }
}

/* ---- paged_attention_v1: csrc/attention/attention_kernels.cuh:75-485 --------------------
 * fp32 logits, exp(l - max) * 1/(sum + 1e-6), probabilities rounded to bf16, fp32 PV. */
void cpu_paged_attention_v1_bf16(bf16* out, const bf16* q, const bf16* kc, const bf16* vc,
                                 int num_seqs, int heads, int kv_heads, int head_size,
                                 int block_size, float scale, const int* block_tables,
                                 const int* seq_lens, int max_blocks, int64_t q_stride,
                                 int64_t kv_block_stride, int64_t kv_head_stride) {
  const int x = 8, G = heads / kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int s = 0; s < num_seqs; ++s) {
    for (int h = 0; h < heads; ++h) {
      const int L = seq_lens[s];
      bf16* o = out + ((size_t)s * heads + h) * head_size;
      if (L == 0) { memset(o, 0, head_size * sizeof(bf16)); continue; }
      float* logits = (float*)malloc((size_t)L * sizeof(float));
      float qf[256];
      for (int d = 0; d < head_size; ++d) qf[d] = bf2f(q[s * q_stride + (int64_t)h * head_size + d]);
      const int kvh = h / G;
      float mx = -INFINITY;
      for (int t = 0; t < L; ++t) {
        const int64_t pb = block_tables[(size_t)s * max_blocks + t / block_size];
        const bf16* kp = kc + pb * kv_block_stride + (int64_t)kvh * kv_head_stride;
        const int off = t % block_size;
        float acc = 0.f;
        for (int d = 0; d < head_size; ++d)
          acc += qf[d] * bf2f(kp[((d / x) * block_size + off) * x + d % x]);
        logits[t] = acc * scale;
        if (logits[t] > mx) mx = logits[t];
      }
      float sum = 0.f;
      for (int t = 0; t < L; ++t) { logits[t] = expf(logits[t] - mx); sum += logits[t]; }
      const float inv = 1.0f / (sum + 1e-6f);
      float acc[256];
      for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
      for (int t = 0; t < L; ++t) {
        const float p = bf2f(f2bf(logits[t] * inv));
        const int64_t pb = block_tables[(size_t)s * max_blocks + t / block_size];
        const bf16* vp = vc + pb * kv_block_stride + (int64_t)kvh * kv_head_stride;
        const int off = t % block_size;
        for (int d = 0; d < head_size; ++d) acc[d] += p * bf2f(vp[(size_t)d * block_size + off]);
      }
      for (int d = 0; d < head_size; ++d) o[d] = f2bf(acc[d]);
      free(logits);
    }
  }
}

/* ---- w4a16 GEMM: csrc/quantization/awq/gemm_kernels.cu:410-463 -> hgemm_gptq.h:2165-2259,
 * dequant w = bf16(fma(q, s, -z*s)) (hgemm_gptq.h:487-570, 869-905); fp32 accumulation.
 * qw: exllama-shuffled words [K/8][N]; zero_mode 0: AWQ-ordered zeros, value z;
 * zero_mode 1: GPTQ natural-order zeros, value z + 1. */
static const int kExl[8] = {0, 2, 4, 6, 1, 3, 5, 7};   /* nibble p -> k row */
static const int kAwqShift[8] = {0, 4, 1, 5, 2, 6, 3, 7}; /* column j -> nibble */

void cpu_w4a16_gemm_bf16(bf16* out, const bf16* x, const uint32_t* qw, const bf16* scales,
                         const uint32_t* qz, int zero_mode, int m, int n, int k, int group,
                         int64_t lda) {
  const int NB = 64;
#pragma omp parallel
  {
    float* wt = (float*)malloc((size_t)group * NB * sizeof(float));
    float* acc = (float*)malloc((size_t)m * NB * sizeof(float));
#pragma omp for schedule(dynamic)
    for (int n0 = 0; n0 < n; n0 += NB) {
      const int nb = n - n0 < NB ? n - n0 : NB;
      memset(acc, 0, (size_t)m * NB * sizeof(float));
      for (int g = 0; g < k / group; ++g) {
        /* dequantise the [group x nb] tile */
        for (int j = 0; j < nb; ++j) {
          const int col = n0 + j;
          const uint32_t zw = qz[(size_t)g * (n / 8) + col / 8];
          const int zi = zero_mode == 0 ? (int)((zw >> (4 * kAwqShift[col % 8])) & 15u)
                                        : (int)((zw >> (4 * (col % 8))) & 15u) + 1;
          const float s = bf2f(scales[(size_t)g * n + col]);
          const float zs = -(float)zi * s;
          for (int kk = 0; kk < group / 8; ++kk) {
            const uint32_t w = qw[((size_t)(g * group / 8 + kk)) * n + col];
            for (int p = 0; p < 8; ++p) {
              const float qv = (float)((w >> (4 * p)) & 15u);
              wt[(size_t)(kk * 8 + kExl[p]) * NB + j] = bf2f(f2bf(fmaf(qv, s, zs)));
            }
          }
        }
        for (int r = 0; r < m; ++r) {
          const bf16* xr = x + (size_t)r * lda + (size_t)g * group;
          float* a = acc + (size_t)r * NB;
          for (int kk = 0; kk < group; ++kk) {
            const float xv = bf2f(xr[kk]);
            const float* wr = wt + (size_t)kk * NB;
            for (int j = 0; j < NB; ++j) a[j] += xv * wr[j];
          }
        }
      }
      for (int r = 0; r < m; ++r)
        for (int j = 0; j < nb; ++j) out[(size_t)r * n + n0 + j] = f2bf(acc[(size_t)r * NB + j]);
    }
    free(wt);
    free(acc);
  }
}

/* ---- un-quantised GEMM for the lm_head (bf16 x bf16 -> fp32 argmax-ready logits) ---------- */
void cpu_gemm_bf16(float* out, const bf16* x, const bf16* w /* [K][N] */, int m, int n, int k) {
#pragma omp parallel for schedule(static)
  for (int n0 = 0; n0 < n; n0 += 64) {
    const int nb = n - n0 < 64 ? n - n0 : 64;
    for (int r = 0; r < m; ++r) {
      float acc[64];
      for (int j = 0; j < 64; ++j) acc[j] = 0.f;
      for (int kk = 0; kk < k; ++kk) {
        const float xv = bf2f(x[(size_t)r * k + kk]);
        const bf16* wr = w + (size_t)kk * n + n0;
        for (int j = 0; j < nb; ++j) acc[j] += xv * bf2f(wr[j]);
      }
      for (int j = 0; j < nb; ++j) out[(size_t)r * n + n0 + j] = acc[j];
    }
  }
}
