// cache_kernels.hip — KV-cache scatter / block-copy kernels for gfx950.
//
// Reference behaviour restated (bit-exact copies):
//   reshape_and_cache        csrc/cache_kernels.cu:203-255, 407-433
//   reshape_and_cache_flash  csrc/cache_kernels.cu:271-344, 450-488
//   copy_blocks              csrc/cache_kernels.cu:65-91, 116-163
//   swap_blocks              csrc/cache_kernels.cu:18-60
//   fp8 (e4m3fn) KV cache     csrc/cache_kernels.cu:245-253, 258-269 (the scaled_convert hook of the
//                             cache write), :544-612 (convert_fp8); the reference's own dispatch
//                             rejects it (quant_utils.cuh:29-42), semantics are upstream vLLM's:
//                             cache byte = sat_e4m3(float(x) / scale), value = T(float(byte) * scale)
//
// MI355X design: all HBM traffic is 16-byte per lane.  The reference's V scatter
// (2-byte stores strided by block_size) is replaced by a 16-token tile that is
// transposed through LDS so that tokens of one KV block land as contiguous runs.
#include "common.cuh"

namespace mi355x {

constexpr int kTokTile = 16;

// ---------------------------------------------------------------------------
// Tiled path: grid (ceil(T/16), num_heads), 256 threads.
// Requires x == 16/sizeof(T), head_size % x == 0, 16-byte aligned rows.
// ROPE (mi355x_rotary_reshape_and_cache): the key rows are rotated (NeoX, rot_dim == head_size, the arithmetic of
// rotary_embedding) on their way into the cache; `key` itself is not modified.
template <typename T, bool ROPE = false>
__global__ __launch_bounds__(256) void reshape_and_cache_tiled_kernel(
    const T* __restrict__ key, const T* __restrict__ value, T* __restrict__ key_cache,
    T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    int num_tokens, int64_t key_stride, int64_t value_stride, int num_heads,
    int head_size, int block_size, const int64_t* __restrict__ positions = nullptr,
    const T* __restrict__ cos_sin_cache = nullptr) {
  constexpr int X = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // V tile [kTokTile][head_size + X] (one 16-B pad per row), then slot tables.
  T* vt = reinterpret_cast<T*>(smem_raw);
  const int row = head_size + X;
  int64_t* s_blk = reinterpret_cast<int64_t*>(smem_raw + (size_t)kTokTile * row * sizeof(T));
  int* s_off = reinterpret_cast<int*>(s_blk + kTokTile);

  const int t0 = blockIdx.x * kTokTile;
  const int head = blockIdx.y;
  const int tid = threadIdx.x;
  const int chunks = head_size / X;  // 16-B chunks per (token, head)

  if (tid < kTokTile) {
    const int t = t0 + tid;
    int64_t slot = (t < num_tokens) ? slot_mapping[t] : -1;
    if (slot < 0) {
      s_blk[tid] = -1;
      s_off[tid] = 0;
    } else {
      s_blk[tid] = slot / block_size;
      s_off[tid] = static_cast<int>(slot % block_size);
    }
  }
  __syncthreads();

  // K: straight 16-B copies; V: stage into LDS.
  for (int i = tid; i < kTokTile * chunks; i += blockDim.x) {
    const int j = i / chunks;
    const int c = i - j * chunks;
    const int64_t blk = s_blk[j];
    if (blk < 0) continue;
    const int t = t0 + j;
    if constexpr (ROPE) {
      // a thread with c < chunks / 2 rotates the chunk pair (c, c + chunks / 2): elements d and d + head_size / 2
      if (c < chunks / 2) {
        const int half = chunks / 2;
        const T* kp = key + t * key_stride + (int64_t)head * head_size;
        Vec16<T> x = load16(kp + c * X);
        Vec16<T> y = load16(kp + (c + half) * X);
        const T* cs_row = cos_sin_cache + positions[t] * head_size;
        const Vec16<T> cs = load16(cs_row + c * X);
        const Vec16<T> sn = load16(cs_row + (c + half) * X);
#pragma unroll
        for (int e = 0; e < X; ++e) rot_pair<T>(x.e[e], y.e[e], cs.e[e], sn.e[e]);
        T* kb = key_cache + ((blk * num_heads + head) * chunks * block_size + s_off[j]) * X;
        store16(kb + (int64_t)c * block_size * X, x);
        store16(kb + (int64_t)(c + half) * block_size * X, y);
      }
    } else {
    const uint4 kv = *reinterpret_cast<const uint4*>(key + t * key_stride +
                                                     (int64_t)head * head_size + c * X);
    T* kdst = key_cache +
              (((blk * num_heads + head) * chunks + c) * block_size + s_off[j]) * X;
    *reinterpret_cast<uint4*>(kdst) = kv;
    }
    const uint4 vv = *reinterpret_cast<const uint4*>(value + t * value_stride +
                                                     (int64_t)head * head_size + c * X);
    *reinterpret_cast<uint4*>(vt + j * row + c * X) = vv;
  }
  __syncthreads();

  // V: transposed write-out, token index fastest across lanes.
  for (int i = tid; i < kTokTile * head_size; i += blockDim.x) {
    const int j = i % kTokTile;
    const int d = i / kTokTile;
    const int64_t blk = s_blk[j];
    if (blk < 0) continue;
    value_cache[((blk * num_heads + head) * head_size + d) * block_size + s_off[j]] =
        vt[j * row + d];
  }
}

// Generic path (any x, any alignment): one workgroup per token, scalar copies.
template <typename T>
__global__ void reshape_and_cache_generic_kernel(
    const T* __restrict__ key, const T* __restrict__ value, T* __restrict__ key_cache,
    T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    int64_t key_stride, int64_t value_stride, int num_heads, int head_size,
    int block_size, int x) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size;
    const int ho = i % head_size;
    const int xi = ho / x;
    const int xo = ho % x;
    const int64_t kdst =
        (((blk * num_heads + head) * (head_size / x) + xi) * block_size + off) * x + xo;
    const int64_t vdst = ((blk * num_heads + head) * head_size + ho) * block_size + off;
    key_cache[kdst] = key[token * key_stride + i];
    value_cache[vdst] = value[token * value_stride + i];
  }
}

// ---------------------------------------------------------------------------
// Flash (NHD / HND) layout: one workgroup per token; 16-B chunks when VEC.
template <typename T, bool VEC>
__global__ void reshape_and_cache_flash_kernel(
    const T* __restrict__ key, const T* __restrict__ value, T* __restrict__ key_cache,
    T* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping,
    int64_t block_stride, int64_t page_stride, int64_t head_stride, int64_t key_stride,
    int64_t value_stride, int num_heads, int head_size, int block_size) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const T* ksrc = key + token * key_stride;
  const T* vsrc = value + token * value_stride;
  T* kdst = key_cache + blk * block_stride + off * page_stride;
  T* vdst = value_cache + blk * block_stride + off * page_stride;
  if constexpr (VEC) {
    constexpr int X = 16 / sizeof(T);
    const int chunks = head_size / X;
    for (int i = threadIdx.x; i < num_heads * chunks; i += blockDim.x) {
      const int head = i / chunks;
      const int c = i - head * chunks;
      const int so = head * head_size + c * X;
      const int64_t dof = (int64_t)head * head_stride + c * X;
      *reinterpret_cast<uint4*>(kdst + dof) = *reinterpret_cast<const uint4*>(ksrc + so);
      *reinterpret_cast<uint4*>(vdst + dof) = *reinterpret_cast<const uint4*>(vsrc + so);
    }
  } else {
    for (int i = threadIdx.x; i < num_heads * head_size; i += blockDim.x) {
      const int head = i / head_size;
      const int d = i - head * head_size;
      const int64_t dof = (int64_t)head * head_stride + d;
      kdst[dof] = ksrc[i];
      vdst[dof] = vsrc[i];
    }
  }
}


// ---------------------------------------------------------------------------
// fp8 (e4m3fn) cache.  key_cache [nb, heads, d/16, bs, 16] bytes (x = 16 / sizeof(cache_t) = 16),
// value_cache [nb, heads, d, bs] bytes.  byte = sat_e4m3(float(x) / *scale), RNE.
template <typename T, bool E5M2 = false>
__device__ __forceinline__ uint4 quant16(const T* __restrict__ src, float scale) {
  // 16 consecutive elements -> 16 fp8 bytes
  constexpr int X = 16 / sizeof(T);
  uint32_t w[4];
  float f[16];
#pragma unroll
  for (int i = 0; i < 16 / X; ++i) {
    const Vec16<T> v = load16<T>(src + i * X);
#pragma unroll
    for (int j = 0; j < X; ++j) f[i * X + j] = to_f32(v.e[j]) / scale;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[i] = (uint32_t)Kv8Fmt<E5M2>::to8x2(f[4 * i], f[4 * i + 1]) |
           ((uint32_t)Kv8Fmt<E5M2>::to8x2(f[4 * i + 2], f[4 * i + 3]) << 16);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// Tiled path: grid (ceil(T/16), num_heads), 256 threads; head_size % 16 == 0, 16-B aligned rows.
template <typename T, bool E5M2 = false>
__global__ __launch_bounds__(256) void reshape_and_cache_fp8_tiled_kernel(
    const T* __restrict__ key, const T* __restrict__ value, uint8_t* __restrict__ key_cache,
    uint8_t* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, int num_tokens,
    int64_t key_stride, int64_t value_stride, int num_heads, int head_size, int block_size,
    const float* __restrict__ k_scale, const float* __restrict__ v_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint8_t* vt = reinterpret_cast<uint8_t*>(smem_raw);            // [kTokTile][head_size + 16]
  const int row = head_size + 16;
  int64_t* s_blk = reinterpret_cast<int64_t*>(smem_raw + (size_t)kTokTile * row);
  int* s_off = reinterpret_cast<int*>(s_blk + kTokTile);
  const int t0 = blockIdx.x * kTokTile;
  const int head = blockIdx.y;
  const int tid = threadIdx.x;
  const int chunks = head_size / 16;
  const float ks = *k_scale, vs = *v_scale;
  if (tid < kTokTile) {
    const int t = t0 + tid;
    int64_t slot = (t < num_tokens) ? slot_mapping[t] : -1;
    s_blk[tid] = slot < 0 ? -1 : slot / block_size;
    s_off[tid] = slot < 0 ? 0 : static_cast<int>(slot % block_size);
  }
  __syncthreads();
  for (int i = tid; i < kTokTile * chunks; i += blockDim.x) {
    const int j = i / chunks;
    const int c = i - j * chunks;
    const int64_t blk = s_blk[j];
    if (blk < 0) continue;
    const int t = t0 + j;
    const uint4 kq = quant16<T, E5M2>(key + t * key_stride + (int64_t)head * head_size + c * 16, ks);
    uint8_t* kdst = key_cache + (((blk * num_heads + head) * chunks + c) * block_size + s_off[j]) * 16;
    *reinterpret_cast<uint4*>(kdst) = kq;
    const uint4 vq = quant16<T, E5M2>(value + t * value_stride + (int64_t)head * head_size + c * 16, vs);
    *reinterpret_cast<uint4*>(vt + j * row + c * 16) = vq;
  }
  __syncthreads();
  for (int i = tid; i < kTokTile * head_size; i += blockDim.x) {
    const int j = i % kTokTile;
    const int d = i / kTokTile;
    const int64_t blk = s_blk[j];
    if (blk < 0) continue;
    value_cache[((blk * num_heads + head) * head_size + d) * block_size + s_off[j]] = vt[j * row + d];
  }
}

template <typename T, bool E5M2 = false>
__global__ void reshape_and_cache_fp8_generic_kernel(
    const T* __restrict__ key, const T* __restrict__ value, uint8_t* __restrict__ key_cache,
    uint8_t* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, int64_t key_stride,
    int64_t value_stride, int num_heads, int head_size, int block_size, int x,
    const float* __restrict__ k_scale, const float* __restrict__ v_scale) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const float ks = *k_scale, vs = *v_scale;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size;
    const int ho = i % head_size;
    const int xi = ho / x;
    const int xo = ho % x;
    const int64_t kdst = (((blk * num_heads + head) * (head_size / x) + xi) * block_size + off) * x + xo;
    const int64_t vdst = ((blk * num_heads + head) * head_size + ho) * block_size + off;
    key_cache[kdst] = Kv8Fmt<E5M2>::to8(to_f32(key[token * key_stride + i]) / ks);
    value_cache[vdst] = Kv8Fmt<E5M2>::to8(to_f32(value[token * value_stride + i]) / vs);
  }
}

template <typename T, bool E5M2 = false>
__global__ void reshape_and_cache_flash_fp8_kernel(
    const T* __restrict__ key, const T* __restrict__ value, uint8_t* __restrict__ key_cache,
    uint8_t* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, int64_t block_stride,
    int64_t page_stride, int64_t head_stride, int64_t key_stride, int64_t value_stride, int num_heads,
    int head_size, int block_size, const float* __restrict__ k_scale,
    const float* __restrict__ v_scale) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const float ks = *k_scale, vs = *v_scale;
  const T* ksrc = key + token * key_stride;
  const T* vsrc = value + token * value_stride;
  uint8_t* kdst = key_cache + blk * block_stride + off * page_stride;
  uint8_t* vdst = value_cache + blk * block_stride + off * page_stride;
  for (int i = threadIdx.x; i < num_heads * head_size; i += blockDim.x) {
    const int head = i / head_size;
    const int d = i - head * head_size;
    const int64_t dof = (int64_t)head * head_stride + d;
    kdst[dof] = Kv8Fmt<E5M2>::to8(to_f32(ksrc[i]) / ks);
    vdst[dof] = Kv8Fmt<E5M2>::to8(to_f32(vsrc[i]) / vs);
  }
}

// convert_fp8 (csrc/cache_kernels.cu:544-612, "only for testing" there): elementwise over the flat
// cache, either direction.
template <typename T, bool TO_FP8, bool E5M2 = false>
__global__ void convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, float scale,
                                   int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (TO_FP8) {
      static_cast<uint8_t*>(dst)[i] = Kv8Fmt<E5M2>::to8(to_f32(static_cast<const T*>(src)[i]) / scale);
    } else {
      static_cast<T*>(dst)[i] = from_f32<T>(Kv8Fmt<E5M2>::from8(static_cast<const uint8_t*>(src)[i]) * scale);
    }
  }
}

// ---------------------------------------------------------------------------
// copy_blocks: layer pointers travel in the kernel argument (no H2D upload).
constexpr int kCopyLayersPerLaunch = 64;
struct CopyBlocksPtrs {
  char* key[kCopyLayersPerLaunch];
  char* value[kCopyLayersPerLaunch];
};

template <typename V>
__global__ void copy_blocks_kernel(CopyBlocksPtrs ptrs,
                                   const int64_t* __restrict__ block_mapping,
                                   int64_t units_per_block) {
  const int layer = blockIdx.x;
  const int pair = blockIdx.y;
  const int64_t src = block_mapping[2 * pair] * units_per_block;
  const int64_t dst = block_mapping[2 * pair + 1] * units_per_block;
  V* kc = reinterpret_cast<V*>(ptrs.key[layer]);
  V* vc = reinterpret_cast<V*>(ptrs.value[layer]);
  for (int64_t i = threadIdx.x; i < units_per_block; i += blockDim.x) kc[dst + i] = kc[src + i];
  for (int64_t i = threadIdx.x; i < units_per_block; i += blockDim.x) vc[dst + i] = vc[src + i];
}

static inline bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

}  // namespace mi355x

using namespace mi355x;

extern "C" {

int mi355x_reshape_and_cache(const void* key, const void* value, void* key_cache,
                             void* value_cache, const int64_t* slot_mapping,
                             int num_tokens, int64_t key_stride, int64_t value_stride,
                             int num_heads, int head_size, int block_size, int x,
                             int dtype, int kv_cache_dtype, const float* k_scale,
                             const float* v_scale, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0 && x > 0,
                 MI355X_EINVAL, "reshape_and_cache: bad sizes");
  MI355X_REQUIRE(kv_cache_dtype == MI355X_KV_AUTO || kv_cache_dtype == MI355X_KV_FP8_E4M3 ||
                     kv_cache_dtype == MI355X_KV_FP8_E5M2,
                 MI355X_EUNSUPPORTED, "Unsupported data type of kv cache: id %d", kv_cache_dtype);
  MI355X_REQUIRE(head_size % x == 0, MI355X_EINVAL,
                 "reshape_and_cache: head_size %d not a multiple of x %d", head_size, x);
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(key && value && key_cache && value_cache && slot_mapping, MI355X_EINVAL,
                 "reshape_and_cache: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kv_cache_dtype != MI355X_KV_AUTO) {
    const bool e5m2 = kv_cache_dtype == MI355X_KV_FP8_E5M2;
    MI355X_REQUIRE(k_scale && v_scale, MI355X_EINVAL, "reshape_and_cache: fp8 cache needs k_scale / v_scale");
    return MI355X_DISPATCH_FLOAT(dtype, [&] {
      constexpr int X = 16 / sizeof(scalar_t);
      const scalar_t* k = static_cast<const scalar_t*>(key);
      const scalar_t* v = static_cast<const scalar_t*>(value);
      uint8_t* kc = static_cast<uint8_t*>(key_cache);
      uint8_t* vc = static_cast<uint8_t*>(value_cache);
      const bool vec = (x == 16) && (head_size % 16 == 0) && (key_stride % X == 0) &&
                       (value_stride % X == 0) && aligned16(k) && aligned16(v) && aligned16(kc);
      if (vec) {
        dim3 grid((num_tokens + kTokTile - 1) / kTokTile, num_heads);
        size_t smem = (size_t)kTokTile * (head_size + 16) + kTokTile * (sizeof(int64_t) + sizeof(int));
        if (e5m2)
          hipLaunchKernelGGL((reshape_and_cache_fp8_tiled_kernel<scalar_t, true>), grid, dim3(256), smem, s, k, v,
                             kc, vc, slot_mapping, num_tokens, key_stride, value_stride, num_heads,
                             head_size, block_size, k_scale, v_scale);
        else
          hipLaunchKernelGGL((reshape_and_cache_fp8_tiled_kernel<scalar_t, false>), grid, dim3(256), smem, s, k, v,
                             kc, vc, slot_mapping, num_tokens, key_stride, value_stride, num_heads,
                             head_size, block_size, k_scale, v_scale);
      } else {
        int threads = num_heads * head_size < 512 ? num_heads * head_size : 512;
        threads = ((threads + 63) / 64) * 64;
        if (e5m2)
          hipLaunchKernelGGL((reshape_and_cache_fp8_generic_kernel<scalar_t, true>), dim3(num_tokens),
                             dim3(threads), 0, s, k, v, kc, vc, slot_mapping, key_stride, value_stride,
                             num_heads, head_size, block_size, x, k_scale, v_scale);
        else
          hipLaunchKernelGGL((reshape_and_cache_fp8_generic_kernel<scalar_t, false>), dim3(num_tokens),
                             dim3(threads), 0, s, k, v, kc, vc, slot_mapping, key_stride, value_stride,
                             num_heads, head_size, block_size, x, k_scale, v_scale);
      }
      return check_launch("reshape_and_cache(fp8)");
    });
  }
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    constexpr int X = 16 / sizeof(scalar_t);
    const scalar_t* k = static_cast<const scalar_t*>(key);
    const scalar_t* v = static_cast<const scalar_t*>(value);
    scalar_t* kc = static_cast<scalar_t*>(key_cache);
    scalar_t* vc = static_cast<scalar_t*>(value_cache);
    const bool vec = (x == X) && (key_stride % X == 0) && (value_stride % X == 0) &&
                     aligned16(k) && aligned16(v) && aligned16(kc);
    if (vec) {
      dim3 grid((num_tokens + kTokTile - 1) / kTokTile, num_heads);
      size_t smem = (size_t)kTokTile * (head_size + X) * sizeof(scalar_t) +
                    kTokTile * (sizeof(int64_t) + sizeof(int));
      hipLaunchKernelGGL(reshape_and_cache_tiled_kernel<scalar_t>, grid, dim3(256), smem, s,
                         k, v, kc, vc, slot_mapping, num_tokens, key_stride, value_stride,
                         num_heads, head_size, block_size);
    } else {
      int threads = num_heads * head_size < 512 ? num_heads * head_size : 512;
      threads = ((threads + 63) / 64) * 64;
      hipLaunchKernelGGL(reshape_and_cache_generic_kernel<scalar_t>, dim3(num_tokens),
                         dim3(threads), 0, s, k, v, kc, vc, slot_mapping, key_stride,
                         value_stride, num_heads, head_size, block_size, x);
    }
    return check_launch("reshape_and_cache");
  });
}

// returns 1 (no error) when the fused form does not apply: run rotary_embedding on the key rows, then reshape_and_cache
int mi355x_rotary_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                                    const int64_t* slot_mapping, const int64_t* positions,
                                    const void* cos_sin_cache, int num_tokens, int64_t key_stride,
                                    int64_t value_stride, int num_heads, int head_size, int block_size, int x,
                                    int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0 && x > 0, MI355X_EINVAL,
                 "rotary_reshape_and_cache: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(key && value && key_cache && value_cache && slot_mapping && positions && cos_sin_cache,
                 MI355X_EINVAL, "rotary_reshape_and_cache: null pointer");
  if (dtype != MI355X_BF16 && dtype != MI355X_F16) return 1;
  if (x != 8 || head_size % 16 != 0 || key_stride % 8 != 0 || value_stride % 8 != 0 || !aligned16(key) ||
      !aligned16(value) || !aligned16(key_cache) || !aligned16(cos_sin_cache))
    return 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_HALF(dtype, [&] {
    dim3 grid((num_tokens + kTokTile - 1) / kTokTile, num_heads);
    size_t smem = (size_t)kTokTile * (head_size + 8) * sizeof(scalar_t) + kTokTile * (sizeof(int64_t) + sizeof(int));
    hipLaunchKernelGGL((reshape_and_cache_tiled_kernel<scalar_t, true>), grid, dim3(256), smem, s,
                       static_cast<const scalar_t*>(key), static_cast<const scalar_t*>(value),
                       static_cast<scalar_t*>(key_cache), static_cast<scalar_t*>(value_cache), slot_mapping,
                       num_tokens, key_stride, value_stride, num_heads, head_size, block_size, positions,
                       static_cast<const scalar_t*>(cos_sin_cache));
    return check_launch("rotary_reshape_and_cache");
  });
}

int mi355x_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                   void* value_cache, const int64_t* slot_mapping,
                                   int num_tokens, int64_t block_stride,
                                   int64_t page_stride, int64_t head_stride,
                                   int64_t key_stride, int64_t value_stride,
                                   int num_heads, int head_size, int block_size,
                                   int dtype, int kv_cache_dtype, const float* k_scale,
                                   const float* v_scale, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0,
                 MI355X_EINVAL, "reshape_and_cache_flash: bad sizes");
  MI355X_REQUIRE(kv_cache_dtype == MI355X_KV_AUTO || kv_cache_dtype == MI355X_KV_FP8_E4M3 ||
                     kv_cache_dtype == MI355X_KV_FP8_E5M2,
                 MI355X_EUNSUPPORTED, "Unsupported data type of kv cache: id %d", kv_cache_dtype);
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(key && value && key_cache && value_cache && slot_mapping, MI355X_EINVAL,
                 "reshape_and_cache_flash: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kv_cache_dtype != MI355X_KV_AUTO) {
    const bool e5m2 = kv_cache_dtype == MI355X_KV_FP8_E5M2;
    MI355X_REQUIRE(k_scale && v_scale, MI355X_EINVAL,
                   "reshape_and_cache_flash: fp8 cache needs k_scale / v_scale");
    return MI355X_DISPATCH_FLOAT(dtype, [&] {
      int work = num_heads * head_size;
      int threads = work < 256 ? ((work + 63) / 64) * 64 : 256;
      if (e5m2)
        hipLaunchKernelGGL((reshape_and_cache_flash_fp8_kernel<scalar_t, true>), dim3(num_tokens), dim3(threads),
                           0, s, static_cast<const scalar_t*>(key), static_cast<const scalar_t*>(value),
                           static_cast<uint8_t*>(key_cache), static_cast<uint8_t*>(value_cache),
                           slot_mapping, block_stride, page_stride, head_stride, key_stride,
                           value_stride, num_heads, head_size, block_size, k_scale, v_scale);
      else
        hipLaunchKernelGGL((reshape_and_cache_flash_fp8_kernel<scalar_t, false>), dim3(num_tokens), dim3(threads),
                           0, s, static_cast<const scalar_t*>(key), static_cast<const scalar_t*>(value),
                           static_cast<uint8_t*>(key_cache), static_cast<uint8_t*>(value_cache),
                           slot_mapping, block_stride, page_stride, head_stride, key_stride,
                           value_stride, num_heads, head_size, block_size, k_scale, v_scale);
      return check_launch("reshape_and_cache_flash(fp8)");
    });
  }
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    constexpr int X = 16 / sizeof(scalar_t);
    const scalar_t* k = static_cast<const scalar_t*>(key);
    const scalar_t* v = static_cast<const scalar_t*>(value);
    scalar_t* kc = static_cast<scalar_t*>(key_cache);
    scalar_t* vc = static_cast<scalar_t*>(value_cache);
    const bool vec = (head_size % X == 0) && (key_stride % X == 0) &&
                     (value_stride % X == 0) && (block_stride % X == 0) &&
                     (page_stride % X == 0) && (head_stride % X == 0) && aligned16(k) &&
                     aligned16(v) && aligned16(kc) && aligned16(vc);
    int work = vec ? num_heads * head_size / X : num_heads * head_size;
    int threads = work < 256 ? ((work + 63) / 64) * 64 : 256;
    if (vec) {
      hipLaunchKernelGGL((reshape_and_cache_flash_kernel<scalar_t, true>), dim3(num_tokens),
                         dim3(threads), 0, s, k, v, kc, vc, slot_mapping, block_stride,
                         page_stride, head_stride, key_stride, value_stride, num_heads,
                         head_size, block_size);
    } else {
      hipLaunchKernelGGL((reshape_and_cache_flash_kernel<scalar_t, false>),
                         dim3(num_tokens), dim3(threads), 0, s, k, v, kc, vc, slot_mapping,
                         block_stride, page_stride, head_stride, key_stride, value_stride,
                         num_heads, head_size, block_size);
    }
    return check_launch("reshape_and_cache_flash");
  });
}

int mi355x_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int to_fp8,
                       int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(numel >= 0, MI355X_EINVAL, "convert_fp8: bad size");
  if (numel == 0) return MI355X_OK;
  MI355X_REQUIRE(dst && src, MI355X_EINVAL, "convert_fp8: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = (int)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
  return MI355X_DISPATCH_FLOAT(dtype, [&] {
    // to_fp8: 0 = e4m3 bytes -> T, 1 = T -> e4m3 bytes, 2 = e5m2 bytes -> T, 3 = T -> e5m2 bytes (2 / 3: round 3)
    if (to_fp8 == 1) {
      hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, true>), dim3(blocks), dim3(256), 0, s, dst, src,
                         scale, numel);
    } else if (to_fp8 == 0) {
      hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, false>), dim3(blocks), dim3(256), 0, s, dst, src,
                         scale, numel);
    } else if (to_fp8 == 3) {
      hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, true, true>), dim3(blocks), dim3(256), 0, s, dst, src,
                         scale, numel);
    } else {
      hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, false, true>), dim3(blocks), dim3(256), 0, s, dst, src,
                         scale, numel);
    }
    return check_launch("convert_fp8");
  });
}

int mi355x_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                       int num_layers, const int64_t* block_mapping, int num_pairs,
                       int64_t bytes_per_block, mi355x_stream stream) {
  MI355X_REQUIRE(num_layers >= 0 && num_pairs >= 0 && bytes_per_block >= 0, MI355X_EINVAL,
                 "copy_blocks: bad sizes");
  if (num_layers == 0 || num_pairs == 0 || bytes_per_block == 0) return MI355X_OK;
  MI355X_REQUIRE(key_cache_ptrs && value_cache_ptrs && block_mapping, MI355X_EINVAL,
                 "copy_blocks: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int base = 0; base < num_layers; base += kCopyLayersPerLaunch) {
    const int nl = (num_layers - base) < kCopyLayersPerLaunch ? (num_layers - base)
                                                               : kCopyLayersPerLaunch;
    CopyBlocksPtrs ptrs;
    bool vec = (bytes_per_block % 16 == 0);
    for (int i = 0; i < kCopyLayersPerLaunch; ++i) {
      ptrs.key[i] = i < nl ? static_cast<char*>(key_cache_ptrs[base + i]) : nullptr;
      ptrs.value[i] = i < nl ? static_cast<char*>(value_cache_ptrs[base + i]) : nullptr;
      if (i < nl) {
        MI355X_REQUIRE(ptrs.key[i] && ptrs.value[i], MI355X_EINVAL,
                       "copy_blocks: null cache pointer for layer %d", base + i);
        vec = vec && aligned16(ptrs.key[i]) && aligned16(ptrs.value[i]);
      }
    }
    dim3 grid(nl, num_pairs);
    if (vec) {
      hipLaunchKernelGGL(copy_blocks_kernel<uint4>, grid, dim3(256), 0, s, ptrs,
                         block_mapping, bytes_per_block / 16);
    } else if (bytes_per_block % 2 == 0) {
      hipLaunchKernelGGL(copy_blocks_kernel<uint16_t>, grid, dim3(256), 0, s, ptrs,
                         block_mapping, bytes_per_block / 2);
    } else {
      hipLaunchKernelGGL(copy_blocks_kernel<uint8_t>, grid, dim3(256), 0, s, ptrs,
                         block_mapping, bytes_per_block);
    }
    int rc = check_launch("copy_blocks");
    if (rc) return rc;
  }
  return MI355X_OK;
}

int mi355x_swap_blocks(const void* src, void* dst, const int64_t* block_mapping,
                       int num_pairs, int64_t block_size_in_bytes, int kind,
                       mi355x_stream stream) {
  MI355X_REQUIRE(num_pairs >= 0 && block_size_in_bytes >= 0, MI355X_EINVAL,
                 "swap_blocks: bad sizes");
  MI355X_REQUIRE(kind >= 0 && kind <= 2, MI355X_EINVAL,
                 "swap_blocks: Invalid device combination (kind=%d)", kind);
  if (num_pairs == 0 || block_size_in_bytes == 0) return MI355X_OK;
  MI355X_REQUIRE(src && dst && block_mapping, MI355X_EINVAL, "swap_blocks: null pointer");
  static const hipMemcpyKind kinds[3] = {hipMemcpyDeviceToDevice, hipMemcpyDeviceToHost,
                                         hipMemcpyHostToDevice};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* sp = static_cast<const char*>(src);
  char* dp = static_cast<char*>(dst);
  int i = 0;
  while (i < num_pairs) {
    // merge a run of pairs whose src and dst blocks both advance by one
    int run = 1;
    while (i + run < num_pairs &&
           block_mapping[2 * (i + run)] == block_mapping[2 * i] + run &&
           block_mapping[2 * (i + run) + 1] == block_mapping[2 * i + 1] + run) {
      ++run;
    }
    hipError_t e = hipMemcpyAsync(dp + block_mapping[2 * i + 1] * block_size_in_bytes,
                                  sp + block_mapping[2 * i] * block_size_in_bytes,
                                  (size_t)run * block_size_in_bytes, kinds[kind], s);
    if (e != hipSuccess) {
      set_error("swap_blocks: hipMemcpyAsync: %s", hipGetErrorString(e));
      return MI355X_ELAUNCH;
    }
    i += run;
  }
  return MI355X_OK;
}

}  // extern "C"
