// fused_decode.hip — MI355X-side fusions of the decode step (no reference op of their own; each is
// bit-identical to the sequence of reference ops it replaces and is tested against that sequence).
//
// qkv_rope_cache: what follows the qkv projection of a decode step, in ONE launch per layer:
//   [T(sum of the GEMM's split-K slabs)]            (w4a16_sum_slabs_kernel, when sk > 0)
//   rotary_embedding(positions, q, k)  NeoX          (csrc/pos_encoding_kernels.cu:10-34, :37-100)
//   reshape_and_cache(k, v, key_cache, value_cache)  (csrc/cache_kernels.cu:203-255)
// One workgroup per token: the qkv row (<= 16 KiB) is staged in LDS, rotated there, then q goes
// back to the qkv buffer (the attention kernel reads it from there) and k / v are scattered
// into the token's slot of the paged x-split cache.  Decode rows are launch-bound (64 tokens x
// 12 KiB): the three launches cost ~15 us per layer, this one ~5.
#include "common.cuh"

namespace mi355x {

template <typename T>
__global__ __launch_bounds__(256) void qkv_rope_cache_kernel(
    T* __restrict__ qkv, int64_t qkv_stride, const float* __restrict__ slabs, int sk,
    int64_t slab_stride, const int64_t* __restrict__ positions, const T* __restrict__ cos_sin_cache,
    T* __restrict__ key_cache, T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    int num_heads, int num_kv_heads, int head_size, int block_size, int64_t key_block_stride,
    int64_t value_block_stride) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* row = reinterpret_cast<T*>(smem);
  const int64_t token = blockIdx.x;
  const int q_size = num_heads * head_size;
  const int kv_size = num_kv_heads * head_size;
  const int width = q_size + 2 * kv_size;
  T* grow = qkv + token * qkv_stride;

  // 1. the row, as T
  for (int i = threadIdx.x * V; i < width; i += blockDim.x * V) {
    if (sk > 0) {
      float acc[V];
      const float* sp = slabs + token * width + i;
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = sp[j];
      for (int s = 1; s < sk; ++s) {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += sp[(int64_t)s * slab_stride + j];
      }
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
  const int embed = head_size / 2;
  const int cph = embed / V;
  const int64_t pos = positions[token];
  const T* cos_ptr = cos_sin_cache + pos * head_size;
  const T* sin_ptr = cos_ptr + embed;
  for (int i = threadIdx.x; i < (num_heads + num_kv_heads) * cph; i += blockDim.x) {
    const int h = i / cph;
    const int c = i - h * cph;
    T* base = row + h * head_size;        // q heads, then the k heads follow contiguously
    Vec16<T> x = load16(base + c * V);
    Vec16<T> y = load16(base + embed + c * V);
    const Vec16<T> cs = load16(cos_ptr + c * V);
    const Vec16<T> sn = load16(sin_ptr + c * V);
#pragma unroll
    for (int j = 0; j < V; ++j) rot_pair<T>(x.e[j], y.e[j], cs.e[j], sn.e[j]);
    store16(base + c * V, x);
    store16(base + embed + c * V, y);
  }
  __syncthreads();

  // 3. q (and, when the row came from slabs, k and v too) back to the qkv buffer; k / v into the cache
  const int back = sk > 0 ? width : q_size + kv_size;   // v is unchanged when it was read from qkv
  for (int i = threadIdx.x * V; i < back; i += blockDim.x * V) store16(grow + i, load16(row + i));
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int t = (int)(slot - blk * block_size);
  const int cpk = head_size / V;          // 16-byte chunks per head
  const T* krow = row + q_size;
  const T* vrow = krow + kv_size;
  for (int i = threadIdx.x; i < num_kv_heads * cpk; i += blockDim.x) {
    const int h = i / cpk;
    const int c = i - h * cpk;
    T* dst = key_cache + blk * key_block_stride + ((int64_t)(h * cpk + c) * block_size + t) * V;
    store16(dst, load16(krow + h * head_size + c * V));
  }
  for (int i = threadIdx.x; i < kv_size; i += blockDim.x) {
    const int h = i / head_size;
    const int d = i - h * head_size;
    value_cache[blk * value_block_stride + ((int64_t)(h * head_size + d)) * block_size + t] = vrow[i];
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
  MI355X_REQUIRE(x == 8 && head_size % 16 == 0 && qkv_stride % 8 == 0 && width * 2 <= 64 * 1024,
                 MI355X_EUNSUPPORTED,
                 "qkv_rope_cache: needs x == 8, head_size %% 16 == 0, qkv_stride %% 8 == 0, row <= 64 KiB");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  MI355X_REQUIRE(al(qkv) && al(cos_sin_cache) && al(key_cache) && (sk == 0 || (slabs && al(slabs))),
                 MI355X_EINVAL, "qkv_rope_cache: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_HALF(dtype, [&] {
    hipLaunchKernelGGL(qkv_rope_cache_kernel<scalar_t>, dim3(num_tokens), dim3(256),
                       (size_t)width * sizeof(scalar_t), s, static_cast<scalar_t*>(qkv), qkv_stride,
                       sk > 0 ? slabs : nullptr, sk, (int64_t)num_tokens * width, positions,
                       static_cast<const scalar_t*>(cos_sin_cache), static_cast<scalar_t*>(key_cache),
                       static_cast<scalar_t*>(value_cache), slot_mapping, num_heads, num_kv_heads,
                       head_size, block_size, key_block_stride, value_block_stride);
    return check_launch("qkv_rope_cache");
  });
}
