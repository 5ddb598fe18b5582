// merge_attn_states.hip — combine two partial attention results (prefix / suffix of the KV range)
// from their log-sum-exps (section 2.2 of arXiv:2501.01005).  SURVEY §8f rank 2.
//
// Reference restated: csrc/attention/merge_attn_states.cu:15-87 (kernel), :133-172 (launcher):
//   lse == +inf is read as -inf;  m = max(p_lse, s_lse);  p_se = expf(p_lse - m), s_se likewise;
//   out = p_out * (p_se / (p_se + s_se)) + s_out * (s_se / (p_se + s_se))   in fp32, the first
//   product fused into the add (fma), one rounding to scalar_t;  out_lse = logf(p_se + s_se) + m.
//   output [T, H, D] (heads contiguous), lse tensors [H, T] fp32.
// HBM-bound elementwise: one thread per 16 bytes of output; the per-(token, head) scalars are
// recomputed by the D/8 threads of a head (two cached loads + two expf) instead of staged.
#include "common.cuh"

namespace mi355x {

template <typename T>
__global__ __launch_bounds__(256) void merge_attn_states_kernel(
    T* __restrict__ output, float* __restrict__ output_lse, const T* __restrict__ prefix_output,
    const float* __restrict__ prefix_lse, const T* __restrict__ suffix_output,
    const float* __restrict__ suffix_lse, int num_tokens, int num_heads, int head_size) {
  constexpr int V = 16 / sizeof(T);
  const int per_head = head_size / V;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)num_tokens * num_heads * per_head;
  if (idx >= total) return;
  const int64_t th = idx / per_head;            // token * num_heads + head
  const int pack = (int)(idx - th * per_head);
  const int token = (int)(th / num_heads);
  const int head = (int)(th - (int64_t)token * num_heads);

  float p_lse = prefix_lse[(int64_t)head * num_tokens + token];
  float s_lse = suffix_lse[(int64_t)head * num_tokens + token];
  const float ninf = -__builtin_huge_valf();
  p_lse = __builtin_isinf(p_lse) ? ninf : p_lse;
  s_lse = __builtin_isinf(s_lse) ? ninf : s_lse;
  const float max_lse = fmaxf(p_lse, s_lse);
  const float p_se = expf(p_lse - max_lse);
  const float s_se = expf(s_lse - max_lse);
  const float out_se = p_se + s_se;
  const float p_scale = p_se / out_se;
  const float s_scale = s_se / out_se;

  const int64_t off = th * head_size + (int64_t)pack * V;
  const Vec16<T> p = load16(prefix_output + off);
  const Vec16<T> s = load16(suffix_output + off);
  Vec16<T> o;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float t;
    {
#pragma clang fp contract(off)
      t = to_f32(s.e[j]) * s_scale;              // rounded product, as the reference's parenthesis
    }
    o.e[j] = from_f32<T>(fmaf(to_f32(p.e[j]), p_scale, t));
  }
  store16(output + off, o);
  if (output_lse != nullptr && pack == 0) {
    output_lse[(int64_t)head * num_tokens + token] = logf(out_se) + max_lse;
  }
}

}  // namespace mi355x

using namespace mi355x;

extern "C" int mi355x_merge_attn_states(void* output, float* output_lse, const void* prefix_output,
                                        const float* prefix_lse, const void* suffix_output,
                                        const float* suffix_lse, int num_tokens, int num_heads,
                                        int head_size, int dtype, mi355x_stream stream) {
  MI355X_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0, MI355X_EINVAL,
                 "merge_attn_states: bad sizes");
  if (num_tokens == 0) return MI355X_OK;
  MI355X_REQUIRE(output && prefix_output && prefix_lse && suffix_output && suffix_lse, MI355X_EINVAL,
                 "merge_attn_states: null pointer");
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  MI355X_REQUIRE(al(output) && al(prefix_output) && al(suffix_output), MI355X_EINVAL,
                 "merge_attn_states: outputs must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return MI355X_DISPATCH_FLOAT(dtype, [&]() -> int {
    constexpr int V = 16 / sizeof(scalar_t);
    // ref: "headsize must be multiple of pack_size" (merge_attn_states.cu:145-146)
    MI355X_REQUIRE(head_size % V == 0, MI355X_EINVAL,
                   "merge_attn_states: headsize must be multiple of pack_size: %d", V);
    const int64_t total = (int64_t)num_tokens * num_heads * (head_size / V);
    hipLaunchKernelGGL(merge_attn_states_kernel<scalar_t>, dim3((unsigned)((total + 255) / 256)),
                       dim3(256), 0, s, static_cast<scalar_t*>(output), output_lse,
                       static_cast<const scalar_t*>(prefix_output), prefix_lse,
                       static_cast<const scalar_t*>(suffix_output), suffix_lse, num_tokens, num_heads,
                       head_size);
    return check_launch("merge_attn_states");
  });
}
