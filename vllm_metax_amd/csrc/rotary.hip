// rotary.hip — in-place rotary position embedding (GPT-NeoX and GPT-J styles).
//
// Reference restated: csrc/pos_encoding_kernels.cu:10-34 (math), :37-100 (loops),
// :133-213 (launcher).  All arithmetic is in scalar_t with a rounding after every
// multiply and after the add/sub, exactly like the c10 scalar operators:
//   x' = T(T(x*cos) - T(y*sin)),  y' = T(T(y*cos) + T(x*sin))
// One workgroup per token; the NeoX vector path moves 16 B per lane (8 x and the 8
// matching y elements) and reads cos/sin as 16-B vectors.
#include "common.cuh"

namespace mi355x {

template <typename T, bool IS_NEOX, bool VEC>
__global__ void rotary_embedding_kernel(const int64_t* __restrict__ positions,
                                        T* __restrict__ query, T* __restrict__ key,
                                        const T* __restrict__ cos_sin_cache, int rot_dim,
                                        int64_t query_stride, int64_t key_stride,
                                        int64_t head_stride, int num_heads,
                                        int num_kv_heads,
                                        const int64_t* __restrict__ cache_offsets) {
  const int64_t token = blockIdx.x;
  // batched_rotary_embedding (multi-LoRA): the cache row is offsets[token] + position
  // (csrc/pos_encoding_kernels.cu:102-129)
  const int64_t pos = positions[token] + (cache_offsets ? cache_offsets[token] : 0);
  const int embed_dim = rot_dim / 2;
  const T* cos_ptr = cos_sin_cache + pos * rot_dim;
  const T* sin_ptr = cos_ptr + embed_dim;
  const int total_heads = num_heads + (key ? num_kv_heads : 0);

  if constexpr (VEC) {
    constexpr int V = 16 / sizeof(T);
    if constexpr (IS_NEOX) {
      const int cph = embed_dim / V;  // chunks per head
      for (int i = threadIdx.x; i < total_heads * cph; i += blockDim.x) {
        const int h = i / cph;
        const int c = i - h * cph;
        T* base = (h < num_heads) ? query + token * query_stride + h * head_stride
                                  : key + token * key_stride + (h - num_heads) * head_stride;
        Vec16<T> x = load16(base + c * V);
        Vec16<T> y = load16(base + embed_dim + c * V);
        const Vec16<T> cs = load16(cos_ptr + c * V);
        const Vec16<T> sn = load16(sin_ptr + c * V);
#pragma unroll
        for (int j = 0; j < V; ++j) rot_pair<T>(x.e[j], y.e[j], cs.e[j], sn.e[j]);
        store16(base + c * V, x);
        store16(base + embed_dim + c * V, y);
      }
    } else {
      // GPT-J: V consecutive elements = V/2 (x,y) pairs, cos/sin indices c*V/2 ..
      const int cph = rot_dim / V;
      for (int i = threadIdx.x; i < total_heads * cph; i += blockDim.x) {
        const int h = i / cph;
        const int c = i - h * cph;
        T* base = (h < num_heads) ? query + token * query_stride + h * head_stride
                                  : key + token * key_stride + (h - num_heads) * head_stride;
        Vec16<T> v = load16(base + c * V);
#pragma unroll
        for (int j = 0; j < V / 2; ++j) {
          const T cs = cos_ptr[c * (V / 2) + j];
          const T sn = sin_ptr[c * (V / 2) + j];
          rot_pair<T>(v.e[2 * j], v.e[2 * j + 1], cs, sn);
        }
        store16(base + c * V, v);
      }
    }
  } else {
    for (int i = threadIdx.x; i < total_heads * embed_dim; i += blockDim.x) {
      const int h = i / embed_dim;
      const int r = i - h * embed_dim;
      T* base = (h < num_heads) ? query + token * query_stride + h * head_stride
                                : key + token * key_stride + (h - num_heads) * head_stride;
      const int xi = IS_NEOX ? r : 2 * r;
      const int yi = IS_NEOX ? embed_dim + r : 2 * r + 1;
      T x = base[xi];
      T y = base[yi];
      rot_pair<T>(x, y, cos_ptr[r], sin_ptr[r]);
      base[xi] = x;
      base[yi] = y;
    }
  }
}

}  // namespace mi355x

using namespace mi355x;

#define LAUNCH_ROT(NEOX, VECF)                                                            \
  hipLaunchKernelGGL((rotary_embedding_kernel<scalar_t, NEOX, VECF>), grid, block, 0, s,   \
                     positions, q, k, cache, rot_dim, query_stride, key_stride, head_stride, \
                     num_heads, num_kv_heads, cache_offsets)

static int rotary_impl(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                       int num_tokens, int rot_dim, int64_t query_stride, int64_t key_stride,
                       int64_t head_stride, int num_heads, int num_kv_heads, int head_size,
                       int is_neox, int dtype, const int64_t* cache_offsets, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && rot_dim > 0 && rot_dim % 2 == 0 && num_heads > 0 &&
                     num_kv_heads >= 0 && head_size > 0,
                 MI355X_EINVAL, "rotary_embedding: bad sizes");
  MI355X_REQUIRE(rot_dim <= head_size, MI355X_EINVAL,
                 "rotary_embedding: rot_dim %d > head_size %d", rot_dim, head_size);
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(positions && query && cos_sin_cache, MI355X_EINVAL,
                 "rotary_embedding: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    constexpr int V = 16 / sizeof(scalar_t);
    scalar_t* q = static_cast<scalar_t*>(query);
    scalar_t* k = static_cast<scalar_t*>(key);
    const scalar_t* cache = static_cast<const scalar_t*>(cos_sin_cache);
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int embed = rot_dim / 2;
    const bool vec = (is_neox ? embed % V == 0 : rot_dim % V == 0) && query_stride % V == 0 &&
                     head_stride % V == 0 && (!k || key_stride % V == 0) && al(q) &&
                     (!k || al(k)) && al(cache) && (!is_neox || rot_dim % V == 0);
    const int total_heads = num_heads + (k ? num_kv_heads : 0);
    const int work = vec ? total_heads * (is_neox ? embed / V : rot_dim / V) : total_heads * embed;
    int threads = ((work + 63) / 64) * 64;
    if (threads > 512) threads = 512;
    dim3 grid(num_tokens), block(threads);
    if (is_neox) {
      if (vec) LAUNCH_ROT(true, true); else LAUNCH_ROT(true, false);
    } else {
      if (vec) LAUNCH_ROT(false, true); else LAUNCH_ROT(false, false);
    }
    return check_launch("rotary_embedding");
  });
}

extern "C" int mi355x_rotary_embedding(const int64_t* positions, void* query, void* key,
                                       const void* cos_sin_cache, int num_tokens,
                                       int rot_dim, int64_t query_stride,
                                       int64_t key_stride, int64_t head_stride,
                                       int num_heads, int num_kv_heads, int head_size,
                                       int is_neox, int dtype, mi355x_stream stream) {
  return rotary_impl(positions, query, key, cos_sin_cache, num_tokens, rot_dim, query_stride,
                     key_stride, head_stride, num_heads, num_kv_heads, head_size, is_neox, dtype,
                     nullptr, stream);
}

extern "C" int mi355x_batched_rotary_embedding(const int64_t* positions, void* query, void* key,
                                               const void* cos_sin_cache,
                                               const int64_t* cos_sin_cache_offsets, int num_tokens,
                                               int rot_dim, int64_t query_stride, int64_t key_stride,
                                               int64_t head_stride, int num_heads, int num_kv_heads,
                                               int head_size, int is_neox, int dtype,
                                               mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens == 0 || cos_sin_cache_offsets, MI355X_EINVAL,
                 "batched_rotary_embedding: cos_sin_cache_offsets is null");
  return rotary_impl(positions, query, key, cos_sin_cache, num_tokens, rot_dim, query_stride,
                     key_stride, head_stride, num_heads, num_kv_heads, head_size, is_neox, dtype,
                     cos_sin_cache_offsets, stream);
}
