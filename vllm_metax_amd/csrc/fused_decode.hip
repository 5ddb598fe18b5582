// fused_decode.hip — MI355X-side fusions of the decode step (no reference op of their own; each is
// bit-identical to the sequence of reference ops it replaces and is tested against that sequence).
//
// qkv_rope_cache: what follows the qkv projection of a decode step, in ONE launch per layer:
//   [T(sum of the GEMM's split-K slabs)]            (w4a16_sum_slabs_kernel, when sk > 0)
//   rotary_embedding(positions, q, k)  NeoX          (csrc/pos_encoding_kernels.cu:10-34, :37-100)
//   reshape_and_cache(k, v, key_cache, value_cache)  (csrc/cache_kernels.cu:203-255)
// One workgroup per (token, group of ~8 heads of the row's q | k | v head list): its columns are
// staged in LDS, rotated there, then q goes back to the qkv buffer (the attention kernel reads it
// from there) and k / v are scattered into the token's slot of the paged x-split cache.  Decode
// rows are launch-bound (64 tokens x 12 KiB): the three launches cost ~15 us per layer.
#include "common.cuh"

namespace mi355x {

template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_cache_kernel(
    T* __restrict__ qkv, int64_t qkv_stride, const float* __restrict__ slabs, int sk,
    int64_t slab_stride, const int64_t* __restrict__ positions, const T* __restrict__ cos_sin_cache,
    T* __restrict__ key_cache, T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    int num_heads, int num_kv_heads, int head_size, int block_size, int64_t key_block_stride,
    int64_t value_block_stride, int heads_per_wg) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // grid (token, head group): a workgroup owns heads [h0, h1) of the row's q | k | v head list, so
  // a 64-token decode batch is 384 workgroups instead of 64
  T* row = reinterpret_cast<T*>(smem);
  const int64_t token = blockIdx.x;
  const int total_heads = num_heads + 2 * num_kv_heads;
  const int h0 = blockIdx.y * heads_per_wg;
  const int h1 = min(h0 + heads_per_wg, total_heads);
  const int c0 = h0 * head_size;                 // first column of this workgroup
  const int cols = (h1 - h0) * head_size;
  const int width = total_heads * head_size;
  T* grow = qkv + token * qkv_stride + c0;

  // position, slot and this thread's cos / sin chunk are fetched up front: behind the slab sum they would
  // be two more dependent memory round trips of a launch that is nothing but round trips
  const int embed = head_size / 2;
  const int cph = embed / V;
  const int rot_heads = min(h1, num_heads + num_kv_heads) - h0;     // <= 0: v heads only
  const int64_t slot = slot_mapping[token];
  Vec16<T> cs0, sn0;
  const bool pre = rot_heads > 0 && (int)threadIdx.x < rot_heads * cph;
  const T* cos_ptr = nullptr;
  if (rot_heads > 0) {
    cos_ptr = cos_sin_cache + positions[token] * head_size;
    if (pre) {
      const int c = (int)threadIdx.x % cph;
      cs0 = load16(cos_ptr + c * V);
      sn0 = load16(cos_ptr + embed + c * V);
    }
  }

  // 1. the columns, as T
  for (int i = threadIdx.x * V; i < cols; i += blockDim.x * V) {
    if (sk > 0) {
      float acc[V];
      sum_slabs<V>(slabs + token * width + c0 + i, sk, slab_stride, acc);
      Vec16<T> v;
#pragma unroll
      for (int j = 0; j < V; ++j) v.e[j] = from_f32<T>(acc[j]);
      store16(row + i, v);
    } else {
      store16(row + i, load16(grow + i));
    }
  }
  __syncthreads();

  // 2. NeoX rotary on the q and k heads (rot_dim == head_size), in LDS
  if (rot_heads > 0) {
    const T* sin_ptr = cos_ptr + embed;
    for (int i = threadIdx.x; i < rot_heads * cph; i += blockDim.x) {
      const int h = i / cph;
      const int c = i - h * cph;
      T* base = row + h * head_size;
      Vec16<T> x = load16(base + c * V);
      Vec16<T> y = load16(base + embed + c * V);
      const bool first = i == (int)threadIdx.x;                 // (the chunk fetched up front)
      const Vec16<T> cs = first ? cs0 : load16(cos_ptr + c * V);
      const Vec16<T> sn = first ? sn0 : load16(sin_ptr + c * V);
#pragma unroll
      for (int j = 0; j < V; ++j) rot_pair<T>(x.e[j], y.e[j], cs.e[j], sn.e[j]);
      store16(base + c * V, x);
      store16(base + embed + c * V, y);
    }
    __syncthreads();
  }

  // 3. q and k (and, when the row came from slabs, v too) back to the qkv buffer; k / v into the cache
  const int back = (sk > 0 ? h1 : min(h1, num_heads + num_kv_heads)) - h0;   // v unchanged when read from qkv
  for (int i = threadIdx.x * V; i < back * head_size; i += blockDim.x * V) {
    store16(grow + i, load16(row + i));
  }
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int t = (int)(slot - blk * block_size);
  const int cpk = head_size / V;          // 16-byte chunks per head
  const int k_lo = max(h0, num_heads), k_hi = min(h1, num_heads + num_kv_heads);
  for (int i = threadIdx.x; i < (k_hi - k_lo) * cpk; i += blockDim.x) {
    const int hh = i / cpk;
    const int c = i - hh * cpk;
    const int h = k_lo + hh - num_heads;                               // kv head
    T* dst = key_cache + blk * key_block_stride + ((int64_t)(h * cpk + c) * block_size + t) * V;
    store16(dst, load16(row + (k_lo + hh - h0) * head_size + c * V));
  }
  const int v_lo = max(h0, num_heads + num_kv_heads);
  for (int i = threadIdx.x; i < (h1 - v_lo) * head_size; i += blockDim.x) {
    const int hh = i / head_size;
    const int d = i - hh * head_size;
    const int h = v_lo + hh - num_heads - num_kv_heads;                // kv head
    value_cache[blk * value_block_stride + ((int64_t)(h * head_size + d)) * block_size + t] =
        row[(v_lo + hh - h0) * head_size + d];
  }
}

// ---- greedy_advance: what follows the lm_head of a greedy decode step, one launch -----------------
// token = argmax(logits[row]) (torch.argmax order: NaN first, lowest index among equal maxima); position, sequence length += 1; slot of
// the token the next step writes = block_table[row][new position / block_size] * block_size + remainder.
// One workgroup per row: lanes stride over 16-byte chunks of the row, (value, index) pairs reduced with
// max-by-value / min-by-index.
template <typename T>
__global__ __launch_bounds__(1024) void greedy_advance_kernel(
    const T* __restrict__ logits, int64_t logits_stride, int vocab, int64_t* __restrict__ tokens,
    int64_t* __restrict__ positions, int* __restrict__ seq_lens, int64_t* __restrict__ slot_mapping,
    const int* __restrict__ block_tables, int max_num_blocks_per_seq, int block_size) {
  constexpr int V = 16 / sizeof(T);
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  const int row = blockIdx.x;
  const T* lr = logits + (int64_t)row * logits_stride;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  // order of torch.argmax: a NaN beats every number, the lowest index wins among equals (and among NaNs) — a row of
  // NaNs must still yield an index inside the vocabulary (the next step gathers the embedding row with it)
  auto take = [&](float v, int i) {
    const bool vn = v != v, bn = best != best;
    const bool better = vn ? (!bn || i < bidx) : (!bn && (v > best || (v == best && i < bidx)));
    if (better) best = v, bidx = i;
  };
  const bool vec = (reinterpret_cast<uintptr_t>(lr) & 15) == 0;
  const int nvec = vec ? vocab / V : 0;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    const Vec16<T> x = load16(lr + c * V);
#pragma unroll
    for (int j = 0; j < V; ++j) take(to_f32(x.e[j]), c * V + j);
  }
  for (int i = nvec * V + threadIdx.x; i < vocab; i += blockDim.x) take(to_f32(lr[i]), i);
  // wave, then workgroup: (max value, min index among equals)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m, 64);
    const int oi = __shfl_xor(bidx, m, 64);
    take(ov, oi);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_val[wave] = best, s_idx[wave] = bidx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) take(s_val[w], s_idx[w]);
    tokens[row] = bidx < vocab ? bidx : 0;
    const int64_t pos = positions[row] + 1;
    positions[row] = pos;
    seq_lens[row] += 1;
    // (a sequence that just filled its last block has no next slot: the index is clamped, the value unused)
    const int64_t bi = pos / block_size < max_num_blocks_per_seq ? pos / block_size : max_num_blocks_per_seq - 1;
    const int64_t blk = block_tables[(int64_t)row * max_num_blocks_per_seq + bi];
    slot_mapping[row] = blk * block_size + pos % block_size;
  }
}

}  // namespace mi355x

using namespace mi355x;

extern "C" int mi355x_qkv_rope_cache(void* qkv, int64_t qkv_stride, const float* slabs, int sk,
                                     const int64_t* positions, const void* cos_sin_cache,
                                     void* key_cache, void* value_cache,
                                     const int64_t* slot_mapping, int num_tokens, int num_heads,
                                     int num_kv_heads, int head_size, int block_size, int x,
                                     int64_t key_block_stride, int64_t value_block_stride,
                                     int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && num_heads > 0 && num_kv_heads > 0 && head_size > 0 &&
                     block_size > 0 && sk >= 0,
                 MI355X_EINVAL, "qkv_rope_cache: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(qkv && positions && cos_sin_cache && key_cache && value_cache && slot_mapping,
                 MI355X_EINVAL, "qkv_rope_cache: null pointer");
  MI355X_REQUIRE(dtype == MI355X_BF16 || dtype == MI355X_F16, MI355X_EUNSUPPORTED,
                 "qkv_rope_cache: 2-byte dtypes only");
  const int width = (num_heads + 2 * num_kv_heads) * head_size;
  MI355X_REQUIRE(x == 8 && head_size % 16 == 0 && qkv_stride % 8 == 0 && head_size <= 8192,
                 MI355X_EUNSUPPORTED,
                 "qkv_rope_cache: needs x == 8, head_size %% 16 == 0 (<= 8192), qkv_stride %% 8 == 0");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  MI355X_REQUIRE(al(qkv) && al(cos_sin_cache) && al(key_cache) && (sk == 0 || (slabs && al(slabs))),
                 MI355X_EINVAL, "qkv_rope_cache: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_HALF(dtype, [&] {
    // ~1024 columns per workgroup (8 heads of 128), 16 B per thread
    const int total_heads = num_heads + 2 * num_kv_heads;
    int hp = 1024 / head_size;
    hp = hp < 1 ? 1 : (hp > total_heads ? total_heads : hp);
    const int chunks = hp * head_size / 8;
    const int threads = chunks >= 256 ? 256 : (chunks + 63) / 64 * 64;
    hipLaunchKernelGGL(qkv_rope_cache_kernel<scalar_t>, dim3(num_tokens, (total_heads + hp - 1) / hp),
                       dim3(threads), (size_t)hp * head_size * sizeof(scalar_t), s,
                       static_cast<scalar_t*>(qkv), qkv_stride,
                       sk > 0 ? slabs : nullptr, sk, (int64_t)num_tokens * width, positions,
                       static_cast<const scalar_t*>(cos_sin_cache), static_cast<scalar_t*>(key_cache),
                       static_cast<scalar_t*>(value_cache), slot_mapping, num_heads, num_kv_heads,
                       head_size, block_size, key_block_stride, value_block_stride, hp);
    return check_launch("qkv_rope_cache");
  });
}

extern "C" int mi355x_greedy_advance(const void* logits, int64_t logits_stride, int num_seqs, int vocab,
                                     int64_t* tokens, int64_t* positions, int* seq_lens,
                                     int64_t* slot_mapping, const int* block_tables,
                                     int max_num_blocks_per_seq, int block_size, int dtype,
                                     mi355x_stream stream) {
  MI355X_REQUIRE(num_seqs >= 0 && vocab > 0 && block_size > 0 && max_num_blocks_per_seq > 0, MI355X_EINVAL,
                 "greedy_advance: bad sizes");
  if (num_seqs == 0) return MI355X_OK;
  MI355X_REQUIRE(logits && tokens && positions && seq_lens && slot_mapping && block_tables, MI355X_EINVAL,
                 "greedy_advance: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    hipLaunchKernelGGL(greedy_advance_kernel<scalar_t>, dim3(num_seqs), dim3(1024), 0, s,
                       static_cast<const scalar_t*>(logits), logits_stride, vocab, tokens, positions,
                       seq_lens, slot_mapping, block_tables, max_num_blocks_per_seq, block_size);
    return check_launch("greedy_advance");
  });
}
