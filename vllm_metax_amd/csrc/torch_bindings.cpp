// torch_bindings.cpp — thin torch.library adapter over the C-ABI (include/mi355x_hotpath.h).
//
// Registers the same op names, in the same namespaces and with the same schema strings,
// as the reference extension (csrc/torch_bindings.cpp:45-69, 112-113, 154-163, 188-209,
// 215-219, 233-247, 251-256, 270-271, 313-322, 327-345, 379-411, 461-468), so that upstream
// vllm/_custom_ops.py finds torch.ops._C.* / torch.ops._C_cache_ops.* / _C_cuda_utils.*
// unchanged.  Every function only extracts pointers / sizes / strides and the current HIP
// stream and forwards them; a non-zero return becomes a c10::Error (Python RuntimeError),
// like the reference's TORCH_CHECK.  No kernels live here (plain g++, no hipcc).
#include <Python.h>

#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <optional>
#include <string>
#include <vector>

#include "mi355x_hotpath.h"

namespace {

using at::Tensor;

inline int dt(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kHalf: return MI355X_F16;
    case at::kBFloat16: return MI355X_BF16;
    case at::kFloat: return MI355X_F32;
    default: TORCH_CHECK(false, "unsupported dtype ", t.scalar_type());
  }
}

inline void* stream_of(const Tensor& t) {
  return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream());
}

struct Guard {
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA g;
  explicit Guard(const Tensor& t) : g(at::device_of(t)) {}
};

inline void ok(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", mi355x_last_error());
}

template <typename T>
inline const T* cptr(const std::optional<Tensor>& t) {
  return (t.has_value() && t->defined() && t->numel() > 0) ? static_cast<const T*>(t->data_ptr())
                                                            : nullptr;
}

// `str kv_cache_dtype` -> mi355x_kv_cache_dtype.  The reference accepts only "auto"
// (csrc/quantization/fp8/metax/quant_utils.cuh:29-42); "fp8" / "fp8_e4m3" (upstream's names) select
// the e4m3fn cache of SURVEY §8f-3.
inline int kv_dtype(const std::string& s, const Tensor& cache, const Tensor& k_scale,
                    const Tensor& v_scale) {
  if (s == "auto") return MI355X_KV_AUTO;
  TORCH_CHECK(s == "fp8" || s == "fp8_e4m3" || s == "fp8_e5m2", "Unsupported data type of kv cache: ", s);
  TORCH_CHECK(cache.element_size() == 1, "kv_cache_dtype ", s, " needs a 1-byte cache tensor");
  TORCH_CHECK(k_scale.is_cuda() && v_scale.is_cuda() && k_scale.scalar_type() == at::kFloat &&
                  v_scale.scalar_type() == at::kFloat && k_scale.numel() == 1 && v_scale.numel() == 1,
              "k_scale / v_scale must be one float32 element each on the GPU");
  return s == "fp8_e5m2" ? MI355X_KV_FP8_E5M2 : MI355X_KV_FP8_E4M3;
}
inline const float* scale_ptr(int kvd, const Tensor& t) {
  return kvd == MI355X_KV_AUTO ? nullptr : t.data_ptr<float>();
}

// ------------------------------------------------------------------- attention
void paged_attention_v1(Tensor& out, Tensor& query, Tensor& key_cache, Tensor& value_cache,
                        int64_t num_kv_heads, double scale, Tensor& block_tables,
                        Tensor& seq_lens, int64_t block_size, int64_t max_seq_len,
                        const std::optional<Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, Tensor& k_scale, Tensor& v_scale,
                        int64_t tp_rank, int64_t blocksparse_local_blocks,
                        int64_t blocksparse_vert_stride, int64_t blocksparse_block_size,
                        int64_t blocksparse_head_sliding_step) {
  const int kvd = kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale);
  TORCH_CHECK(blocksparse_vert_stride <= 1, "block-sparse paged attention is not supported");
  Guard g(query);
  ok(mi355x_paged_attention_v1(
         out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
         query.size(0), query.size(1), num_kv_heads, query.size(2), block_size, (float)scale,
         block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), block_tables.size(1),
         max_seq_len, cptr<float>(alibi_slopes), query.stride(0), key_cache.stride(0),
         key_cache.stride(1), dt(query), kvd, scale_ptr(kvd, k_scale), scale_ptr(kvd, v_scale),
         stream_of(query)),
     "paged_attention_v1");
}

void paged_attention_v2(Tensor& out, Tensor& exp_sums, Tensor& max_logits, Tensor& tmp_out,
                        Tensor& query, Tensor& key_cache, Tensor& value_cache,
                        int64_t num_kv_heads, double scale, Tensor& block_tables,
                        Tensor& seq_lens, int64_t block_size, int64_t max_seq_len,
                        const std::optional<Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, Tensor& k_scale, Tensor& v_scale,
                        int64_t tp_rank, int64_t blocksparse_local_blocks,
                        int64_t blocksparse_vert_stride, int64_t blocksparse_block_size,
                        int64_t blocksparse_head_sliding_step) {
  const int kvd = kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale);
  TORCH_CHECK(blocksparse_vert_stride <= 1, "block-sparse paged attention is not supported");
  Guard g(query);
  ok(mi355x_paged_attention_v2(
         out.data_ptr(), exp_sums.data_ptr<float>(), max_logits.data_ptr<float>(),
         tmp_out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
         query.size(0), query.size(1), num_kv_heads, query.size(2), block_size, (float)scale,
         block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), block_tables.size(1),
         max_seq_len, cptr<float>(alibi_slopes), query.stride(0), key_cache.stride(0),
         key_cache.stride(1), dt(query), kvd, scale_ptr(kvd, k_scale), scale_ptr(kvd, v_scale),
         stream_of(query)),
     "paged_attention_v2");
}

// ----------------------------------------------------------- merge_attn_states
// ref: csrc/attention/merge_attn_states.cu:133-172 (launcher checks), schema torch_bindings.cpp:74-82
void merge_attn_states(Tensor& output, std::optional<Tensor> output_lse, const Tensor& prefix_output,
                       const Tensor& prefix_lse, const Tensor& suffix_output,
                       const Tensor& suffix_lse) {
  const int64_t head_size = output.size(2);
  TORCH_CHECK(output.stride(-2) == head_size && output.stride(-1) == 1,
              "output heads must be contiguous in memory");
  TORCH_CHECK(prefix_output.stride(-2) == head_size && prefix_output.stride(-1) == 1,
              "prefix_output heads must be contiguous in memory");
  TORCH_CHECK(suffix_output.stride(-2) == head_size && suffix_output.stride(-1) == 1,
              "suffix_output heads must be contiguous in memory");
  Guard g(prefix_output);
  ok(mi355x_merge_attn_states(
         output.data_ptr(), output_lse.has_value() ? output_lse->data_ptr<float>() : nullptr,
         prefix_output.data_ptr(), prefix_lse.data_ptr<float>(), suffix_output.data_ptr(),
         suffix_lse.data_ptr<float>(), output.size(0), output.size(1), head_size, dt(output),
         stream_of(output)),
     "merge_attn_states");
}

// ------------------------------------------------------------------ activation
void silu_and_mul(Tensor& out, Tensor& input) {
  TORCH_CHECK(out.is_contiguous() && input.is_contiguous());
  const int64_t d = input.size(-1) / 2;
  const int64_t tokens = d ? input.numel() / input.size(-1) : 0;
  Guard g(input);
  ok(mi355x_silu_and_mul(out.data_ptr(), input.data_ptr(), tokens, d, dt(input), stream_of(input)),
     "silu_and_mul");
}

// ref: csrc/quantization/activation_kernels.cu:117-127
void silu_and_mul_quant(Tensor& out, Tensor& input, Tensor& scale) {
  TORCH_CHECK(out.dtype() == at::kFloat8_e4m3fn, "silu_and_mul_quant: out must be float8_e4m3fn");
  TORCH_CHECK(input.dtype() == at::kHalf || input.dtype() == at::kBFloat16);
  TORCH_CHECK(input.size(-1) % 2 == 0);
  TORCH_CHECK(scale.dtype() == at::kFloat && scale.numel() == 1);
  TORCH_CHECK(out.is_contiguous() && input.is_contiguous());
  const int64_t d = input.size(-1) / 2;
  const int64_t tokens = d ? input.numel() / input.size(-1) : 0;
  Guard g(input);
  ok(mi355x_silu_and_mul_quant(out.data_ptr(), input.data_ptr(),
                               static_cast<const float*>(scale.data_ptr()), tokens, d, dt(input),
                               stream_of(input)),
     "silu_and_mul_quant");
}

// ------------------------------------------------------------------- layernorm
inline int64_t row_stride(const Tensor& t) { return t.dim() >= 2 ? t.stride(-2) : t.size(-1); }
inline int64_t rows(const Tensor& t) { return t.size(-1) ? t.numel() / t.size(-1) : 0; }

void rms_norm(Tensor& out, Tensor& input, Tensor& weight, double epsilon) {
  TORCH_CHECK(out.is_contiguous());
  TORCH_CHECK(input.stride(-1) == 1);
  TORCH_CHECK(weight.is_contiguous());
  Guard g(input);
  ok(mi355x_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), (float)epsilon,
                     rows(input), input.size(-1), row_stride(input), dt(input), stream_of(input)),
     "rms_norm");
}

void fused_add_rms_norm(Tensor& input, Tensor& residual, Tensor& weight, double epsilon) {
  TORCH_CHECK(residual.is_contiguous());
  TORCH_CHECK(weight.is_contiguous());
  Guard g(input);
  ok(mi355x_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                               (float)epsilon, rows(input), input.size(-1), row_stride(input),
                               dt(input), stream_of(input)),
     "fused_add_rms_norm");
}

inline void check_fp8(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kFloat8_e4m3fn, "expected a float8_e4m3fn tensor");
}

void rms_norm_static_fp8_quant(Tensor& out, Tensor& input, Tensor& weight, Tensor& scale,
                               double epsilon) {
  TORCH_CHECK(out.is_contiguous());
  check_fp8(out);
  Guard g(input);
  ok(mi355x_rms_norm_static_fp8_quant(out.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                      scale.data_ptr<float>(), (float)epsilon, rows(input),
                                      input.size(-1), row_stride(input), dt(input),
                                      stream_of(input)),
     "rms_norm_static_fp8_quant");
}

void fused_add_rms_norm_static_fp8_quant(Tensor& out, Tensor& input, Tensor& residual,
                                         Tensor& weight, Tensor& scale, double epsilon) {
  TORCH_CHECK(out.is_contiguous());
  TORCH_CHECK(residual.is_contiguous());
  check_fp8(out);
  Guard g(input);
  ok(mi355x_fused_add_rms_norm_static_fp8_quant(
         out.data_ptr(), input.data_ptr(), residual.data_ptr(), weight.data_ptr(),
         scale.data_ptr<float>(), (float)epsilon, rows(input), input.size(-1), row_stride(input),
         dt(input), stream_of(input)),
     "fused_add_rms_norm_static_fp8_quant");
}

void rms_norm_dynamic_per_token_quant(Tensor& out, const Tensor& input, const Tensor& weight,
                                      Tensor& scales, double epsilon,
                                      std::optional<Tensor> scale_ub,
                                      std::optional<Tensor> residual) {
  check_fp8(out);  // int8 branch: out of scope
  TORCH_CHECK(out.is_contiguous() && input.is_contiguous());
  TORCH_CHECK(scales.scalar_type() == at::kFloat);
  Guard g(input);
  void* res = (residual.has_value() && residual->defined()) ? residual->data_ptr() : nullptr;
  ok(mi355x_rms_norm_dynamic_per_token_quant(out.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                             scales.data_ptr<float>(), (float)epsilon,
                                             cptr<float>(scale_ub), res, rows(input),
                                             input.size(-1), dt(input), stream_of(input)),
     "rms_norm_dynamic_per_token_quant");
}

// ---------------------------------------------------------------------- rotary
static void rotary_common(Tensor& positions, Tensor& query, std::optional<Tensor>& key,
                          int64_t head_size, Tensor& cos_sin_cache, bool is_neox,
                          const int64_t* offsets) {
  // shape / stride handling as in the reference launchers (pos_encoding_kernels.cu:133-213, :219-306)
  const int64_t num_tokens = positions.numel();
  const int pdim = positions.dim();
  TORCH_CHECK(pdim == 1 || pdim == 2,
              "positions must have shape [num_tokens] or [batch_size, seq_len]");
  const bool has_key = key.has_value() && key->defined();
  if (pdim == 1) {
    TORCH_CHECK(query.size(0) == positions.size(0) && (!has_key || key->size(0) == positions.size(0)),
                "query, key and positions must have the same number of tokens");
  } else {
    TORCH_CHECK(query.size(0) == positions.size(0) && query.size(1) == positions.size(1) &&
                    (!has_key || (key->size(0) == positions.size(0) && key->size(1) == positions.size(1))),
                "query, key and positions must have the same batch_size and seq_len");
  }
  const int64_t q_hidden = query.numel() / num_tokens;
  const int64_t k_hidden = has_key ? key->numel() / num_tokens : 0;
  TORCH_CHECK(q_hidden % head_size == 0 && k_hidden % head_size == 0);
  const int64_t num_heads = q_hidden / head_size;
  const int64_t num_kv_heads = has_key ? k_hidden / head_size : num_heads;
  TORCH_CHECK(num_heads % num_kv_heads == 0);
  const int seq_dim = pdim - 1;
  const int64_t query_stride = query.stride(seq_dim);
  const int64_t key_stride = has_key ? key->stride(seq_dim) : 0;
  const int64_t head_stride = (query.dim() == pdim + 2) ? query.stride(-2) : head_size;
  Guard g(query);
  if (offsets) {
    ok(mi355x_batched_rotary_embedding(positions.data_ptr<int64_t>(), query.data_ptr(),
                                       has_key ? key->data_ptr() : nullptr, cos_sin_cache.data_ptr(),
                                       offsets, num_tokens, cos_sin_cache.size(1), query_stride,
                                       key_stride, head_stride, num_heads, num_kv_heads, head_size,
                                       is_neox ? 1 : 0, dt(query), stream_of(query)),
       "batched_rotary_embedding");
    return;
  }
  ok(mi355x_rotary_embedding(positions.data_ptr<int64_t>(), query.data_ptr(),
                             has_key ? key->data_ptr() : nullptr, cos_sin_cache.data_ptr(),
                             num_tokens, cos_sin_cache.size(1), query_stride, key_stride,
                             head_stride, num_heads, num_kv_heads, head_size, is_neox ? 1 : 0,
                             dt(query), stream_of(query)),
     "rotary_embedding");
}

void rotary_embedding(Tensor& positions, Tensor& query, std::optional<Tensor> key,
                      int64_t head_size, Tensor& cos_sin_cache, bool is_neox) {
  rotary_common(positions, query, key, head_size, cos_sin_cache, is_neox, nullptr);
}

// ref: csrc/pos_encoding_kernels.cu:219-306
void batched_rotary_embedding(Tensor& positions, Tensor& query, std::optional<Tensor> key,
                              int64_t head_size, Tensor& cos_sin_cache, bool is_neox, int64_t rot_dim,
                              Tensor& cos_sin_cache_offsets) {
  TORCH_CHECK(positions.size(0) == cos_sin_cache_offsets.size(0) ||
                  positions.numel() == cos_sin_cache_offsets.size(0),
              "positions must have the same num_tokens or batch_size as cos_sin_cache_offsets");
  TORCH_CHECK(rot_dim == cos_sin_cache.size(1), "rot_dim must equal cos_sin_cache.size(1)");
  TORCH_CHECK(cos_sin_cache_offsets.numel() == positions.numel(),
              "one cache offset per token is expected");
  rotary_common(positions, query, key, head_size, cos_sin_cache, is_neox,
                cos_sin_cache_offsets.data_ptr<int64_t>());
}

// ------------------------------------------------------------- int4 weight-only
Tensor awq_to_gptq_4bit(Tensor qweight) {
  TORCH_CHECK(qweight.scalar_type() == at::kInt && qweight.is_contiguous());
  const int64_t k = qweight.size(0), n = qweight.size(1) * 8;
  Guard g(qweight);
  Tensor out = at::zeros({n, (k + 7) / 8}, qweight.options());  // declared [N, K/8], memory [K/8, N]
  ok(mi355x_awq_to_gptq_4bit(static_cast<uint32_t*>(out.data_ptr()),
                             static_cast<const uint32_t*>(qweight.data_ptr()), k, n,
                             stream_of(qweight)),
     "awq_to_gptq_4bit");
  return out;
}

Tensor awq_dequantize(Tensor kernel, Tensor scaling_factors, Tensor zeros, c10::SymInt split_k_iters,
                      int64_t thx, int64_t thy) {
  const int64_t k = kernel.size(0), n = kernel.size(1) * 8;
  const int64_t group = k / scaling_factors.size(0);
  Guard g(scaling_factors);
  Tensor out = at::empty({k, n}, scaling_factors.options());
  ok(mi355x_awq_dequantize(out.data_ptr(), static_cast<const uint32_t*>(kernel.data_ptr()),
                           scaling_factors.data_ptr(),
                           static_cast<const uint32_t*>(zeros.data_ptr()), k, n, group,
                           dt(scaling_factors), stream_of(scaling_factors)),
     "awq_dequantize");
  return out;
}

Tensor awq_gemm(Tensor in_feats, Tensor kernel, Tensor scaling_factors, Tensor zeros,
                c10::SymInt split_k_iters, Tensor temp_space, bool dtype_bf16) {
  TORCH_CHECK(in_feats.dim() == 2 && in_feats.stride(1) == 1);
  TORCH_CHECK(dtype_bf16 == (in_feats.scalar_type() == at::kBFloat16),
              "awq_gemm: dtype_bf16 does not match the input dtype");
  const int64_t m = in_feats.size(0), k = in_feats.size(1);
  const int64_t n = kernel.size(0);  // declared [N, K/8]
  const int64_t group = k / scaling_factors.size(0);
  Guard g(in_feats);
  Tensor out = at::empty({m, n}, in_feats.options());
  float* ws = nullptr;
  int64_t ws_elems = 0;
  if (temp_space.defined() && temp_space.is_cuda() && temp_space.scalar_type() == at::kFloat &&
      temp_space.numel() > 0) {
    ws = temp_space.data_ptr<float>();
    ws_elems = temp_space.numel();
  }
  Tensor dq;  // scratch for the dequantised weights of prefill-sized GEMMs
  if (m >= 1024)   // operand images + the partial tiles of a K split (shapes with few 256 x 256 tiles)
    dq = at::empty({(n + (m + 15) / 16 * 16) * k * 2 + 4 * mi355x_w4a16_prepacked_split_elems((int)m, (int)n, (int)k)},
                   in_feats.options().dtype(at::kByte));
  else if (m > 64) dq = at::empty({8 * 128 * n * 4}, in_feats.options().dtype(at::kByte));   // split-K slabs of a 128-row pass
  ok(mi355x_awq_gemm(out.data_ptr(), in_feats.data_ptr(),
                     static_cast<const uint32_t*>(kernel.data_ptr()), scaling_factors.data_ptr(),
                     static_cast<const uint32_t*>(zeros.data_ptr()), ws, ws_elems,
                     dq.defined() ? dq.data_ptr() : nullptr, dq.defined() ? dq.numel() : 0, m, n, k,
                     group, in_feats.stride(0), dt(in_feats), stream_of(in_feats)),
     "awq_gemm");
  return out;
}

void gptq_shuffle(Tensor q_weight, Tensor q_perm, int64_t bit) {
  Guard g(q_weight);
  const bool has_perm = q_perm.defined() && !q_perm.is_meta() && q_perm.numel() > 0;
  Tensor scratch, perm;
  if (has_perm) {
    perm = q_perm.to(at::kInt).contiguous();
    scratch = at::empty_like(q_weight);
  }
  ok(mi355x_gptq_shuffle(static_cast<uint32_t*>(q_weight.data_ptr()),
                         has_perm ? perm.data_ptr<int>() : nullptr,
                         has_perm ? static_cast<uint32_t*>(scratch.data_ptr()) : nullptr,
                         q_weight.size(0) * 32 / bit, q_weight.size(1), bit, stream_of(q_weight)),
     "gptq_shuffle");
}

Tensor gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, Tensor b_gptq_scales,
                 Tensor b_g_idx, bool use_exllama, int64_t bit, int64_t group_size,
                 Tensor perm_space, Tensor temp_space, bool dtype_bf16) {
  TORCH_CHECK(use_exllama, "gptq_gemm: only the exllama (shuffled) layout is supported");
  TORCH_CHECK(a.is_contiguous());
  TORCH_CHECK(dtype_bf16 == (a.scalar_type() == at::kBFloat16),
              "gptq_gemm: dtype_bf16 does not match the input dtype");
  const int64_t m = a.size(0), k = a.size(1), n = b_q_weight.size(1);
  Guard g(a);
  Tensor out = at::empty({m, n}, a.options());
  const bool has_idx = b_g_idx.defined() && !b_g_idx.is_meta() && b_g_idx.numel() > 0;
  Tensor idx, pspace;
  if (has_idx) {
    idx = b_g_idx.to(at::kInt).contiguous();
    pspace = (perm_space.defined() && perm_space.is_cuda() && perm_space.numel() >= m * k &&
              perm_space.element_size() == 2)
                 ? perm_space
                 : at::empty({m, k}, a.options());
  }
  float* ws = nullptr;
  int64_t ws_elems = 0;
  if (temp_space.defined() && temp_space.is_cuda() && temp_space.scalar_type() == at::kFloat &&
      temp_space.numel() > 0) {
    ws = temp_space.data_ptr<float>();
    ws_elems = temp_space.numel();
  }
  Tensor dq;
  if (m >= 1024)   // operand images + the partial tiles of a K split (shapes with few 256 x 256 tiles)
    dq = at::empty({(n + (m + 15) / 16 * 16) * k * 2 + 4 * mi355x_w4a16_prepacked_split_elems((int)m, (int)n, (int)k)},
                   a.options().dtype(at::kByte));
  else if (m > 64) dq = at::empty({8 * 128 * n * 4}, a.options().dtype(at::kByte));   // split-K slabs of a 128-row pass
  ok(mi355x_gptq_gemm(out.data_ptr(), a.data_ptr(),
                      static_cast<const uint32_t*>(b_q_weight.data_ptr()),
                      static_cast<const uint32_t*>(b_gptq_qzeros.data_ptr()),
                      b_gptq_scales.data_ptr(), has_idx ? idx.data_ptr<int>() : nullptr,
                      has_idx ? pspace.data_ptr() : nullptr, ws, ws_elems,
                      dq.defined() ? dq.data_ptr() : nullptr, dq.defined() ? dq.numel() : 0, m, n,
                      k, bit, group_size, dt(a), stream_of(a)),
     "gptq_gemm");
  return out;
}

// -------------------------------------------------------------------- fp8 GEMM
void cutlass_scaled_mm(Tensor& out, const Tensor& a, const Tensor& b, const Tensor& a_scales,
                       const Tensor& b_scales, const std::optional<Tensor>& bias) {
  // checks as in the reference entry (scaled_mm_entry.cu:84-140)
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && out.dim() == 2);
  TORCH_CHECK(out.size(0) == a.size(0) && a.size(1) == b.size(0) && b.size(1) == out.size(1));
  TORCH_CHECK(a.stride(1) == 1 && out.stride(1) == 1);  // row-major
  TORCH_CHECK(b.stride(0) == 1);                         // column-major
  TORCH_CHECK(out.stride(0) % 16 == 0 && b.stride(1) % 16 == 0);
  const bool is_i8 = a.scalar_type() == at::kChar;
  TORCH_CHECK(a.scalar_type() == b.scalar_type() && (is_i8 || a.scalar_type() == at::kFloat8_e4m3fn),
              "cutlass_scaled_mm: a and b must both be int8 or both float8_e4m3fn");
  const int64_t m = a.size(0), k = a.size(1), n = b.size(1);
  TORCH_CHECK(a_scales.numel() == 1 || a_scales.numel() == m);
  TORCH_CHECK(b_scales.numel() == 1 || b_scales.numel() == n);
  TORCH_CHECK(a_scales.is_contiguous() && b_scales.is_contiguous());
  TORCH_CHECK(a_scales.scalar_type() == at::kFloat && b_scales.scalar_type() == at::kFloat);
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->numel() == n && bias->is_contiguous() && bias->dim() == 1 &&
                bias->scalar_type() == out.scalar_type());
  }
  Guard g(a);
  Tensor ws;  // small-M (decode) shapes split K across up to 8 workgroups, one 4-byte partial slab [m, n] each
  // (64 < m <= 320: passes of 64 rows through the decode kernel; above: the packed-image kernel)
  if (m <= 320 && m > 0) ws = at::empty({8, m < 64 ? m : 64, n}, a.options().dtype(at::kFloat));
  else if (m > 320 && k % 64 == 0) {
    // scratch for the re-tiled operands (bytes / 4) — none for the operands the GEMM reads in place (fp8_gemm.hip
    // run_fp8: fp8, k % 128 == 0, 16-byte aligned rows; MI355X_F8_ROWMAJOR for A/B runs)
    const char* e = getenv("MI355X_F8_ROWMAJOR");
    const int bits = e ? atoi(e) : 3;
    const bool wide = !is_i8 && k % 128 == 0;
    const bool a_in_place = wide && (bits & 1) && a.stride(0) % 16 == 0 && reinterpret_cast<uintptr_t>(a.data_ptr()) % 16 == 0;
    const bool b_in_place = a_in_place && (bits & 2) && b.stride(1) % 16 == 0 &&
                            reinterpret_cast<uintptr_t>(b.data_ptr()) % 16 == 0;
    const int64_t need = (a_in_place ? 0 : (m + 15) / 16 * 16 * k) + (b_in_place ? 0 : (n + 15) / 16 * 16 * k);
    // + the partial tiles of a K split (shapes with few 256 x 256 tiles)
    const int64_t elems = (need + 15) / 16 * 4 + mi355x_scaled_mm_split_elems((int)m, (int)n, (int)k);
    if (elems) ws = at::empty({elems}, a.options().dtype(at::kFloat));
  }
  auto fn = is_i8 ? mi355x_scaled_mm_int8 : mi355x_scaled_mm_fp8;   // scaled_mm_entry.cu:34-39 / new
  ok(fn(out.data_ptr(), a.data_ptr(), b.data_ptr(), a_scales.data_ptr<float>(), a_scales.numel(),
        b_scales.data_ptr<float>(), b_scales.numel(),
        (bias.has_value() && bias->defined()) ? bias->data_ptr() : nullptr,
        ws.defined() ? ws.data_ptr<float>() : nullptr, ws.defined() ? ws.numel() : 0, m, n, k,
        a.stride(0), b.stride(1), out.stride(0), dt(out), stream_of(a)),
     "cutlass_scaled_mm");
}

bool cutlass_scaled_mm_supports_fp8(int64_t cuda_device_capability) { return true; }

// ------------------------------------------------------------------ int8 quant
// ref: csrc/quantization/compressed_tensors/int8_quant_kernels.cu (launchers), schema
// torch_bindings.cpp:349-360.  Symmetric only: azp must be absent.
void static_scaled_int8_quant(Tensor& out, const Tensor& input, const Tensor& scale,
                              const std::optional<Tensor>& azp) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous());
  TORCH_CHECK(scale.numel() == 1);
  TORCH_CHECK(!azp.has_value() || !azp->defined(), "static_scaled_int8_quant: azp is not implemented");
  Guard g(input);
  ok(mi355x_static_scaled_int8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(),
                                     rows(input), input.size(-1), input.size(-1), dt(input),
                                     stream_of(input)),
     "static_scaled_int8_quant");
}

void dynamic_scaled_int8_quant(Tensor& out, const Tensor& input, Tensor& scales,
                               const std::optional<Tensor>& azp) {
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous() && scales.is_contiguous());
  TORCH_CHECK(!azp.has_value() || !azp->defined(), "dynamic_scaled_int8_quant: azp is not implemented");
  Guard g(input);
  ok(mi355x_dynamic_scaled_int8_quant(out.data_ptr(), input.data_ptr(), scales.data_ptr<float>(),
                                      rows(input), input.size(-1), input.size(-1), dt(input),
                                      stream_of(input)),
     "dynamic_scaled_int8_quant");
}

// ------------------------------------------------------------------- fp8 quant
void static_scaled_fp8_quant(Tensor& out, const Tensor& input, const Tensor& scale) {
  TORCH_CHECK(input.stride(-1) == 1 && out.stride(-1) == 1, "last dimension must be contiguous");
  check_fp8(out);
  Guard g(input);
  ok(mi355x_static_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(),
                                    rows(input), input.size(-1), row_stride(input),
                                    row_stride(out), dt(input), stream_of(input)),
     "static_scaled_fp8_quant");
}

void dynamic_scaled_fp8_quant(Tensor& out, const Tensor& input, Tensor& scale) {
  TORCH_CHECK(input.stride(-1) == 1 && out.stride(-1) == 1, "last dimension must be contiguous");
  check_fp8(out);
  Guard g(input);
  ok(mi355x_dynamic_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(),
                                     rows(input), input.size(-1), row_stride(input),
                                     row_stride(out), dt(input), stream_of(input)),
     "dynamic_scaled_fp8_quant");
}

void dynamic_per_token_scaled_fp8_quant(Tensor& out, const Tensor& input, Tensor& scales,
                                        const std::optional<Tensor>& scale_ub) {
  TORCH_CHECK(input.stride(-1) == 1 && out.stride(-1) == 1, "last dimension must be contiguous");
  check_fp8(out);
  Guard g(input);
  ok(mi355x_dynamic_per_token_scaled_fp8_quant(out.data_ptr(), input.data_ptr(),
                                               scales.data_ptr<float>(), cptr<float>(scale_ub),
                                               rows(input), input.size(-1), row_stride(input),
                                               row_stride(out), dt(input), stream_of(input)),
     "dynamic_per_token_scaled_fp8_quant");
}

// ------------------------------------------------------------------- cache ops
void swap_blocks(Tensor& src, Tensor& dst, const Tensor& block_mapping) {
  int kind;
  if (src.is_cuda() && dst.is_cuda()) {
    TORCH_CHECK(src.get_device() == dst.get_device(), "src and dst must be on the same GPU");
    kind = 0;
  } else if (src.is_cuda() && dst.is_cpu()) {
    kind = 1;
  } else if (src.is_cpu() && dst.is_cuda()) {
    kind = 2;
  } else {
    TORCH_CHECK(false, "Invalid device combination");
  }
  TORCH_CHECK(block_mapping.device().is_cpu(), "block_mapping must be on CPU");
  Tensor bm = block_mapping.to(at::kLong).contiguous();
  const Tensor& dev_t = src.is_cuda() ? src : dst;
  Guard g(dev_t);
  ok(mi355x_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr<int64_t>(), bm.size(0),
                        src.element_size() * src.stride(0), kind, stream_of(dev_t)),
     "swap_blocks");
}

void copy_blocks(std::vector<Tensor> const& key_caches, std::vector<Tensor> const& value_caches,
                 const Tensor& block_mapping) {
  const int64_t num_layers = key_caches.size();
  TORCH_CHECK(num_layers == (int64_t)value_caches.size());
  if (num_layers == 0) return;
  TORCH_CHECK(key_caches[0].is_cuda());
  std::vector<void*> kp(num_layers), vp(num_layers);
  for (int64_t i = 0; i < num_layers; ++i) {
    kp[i] = key_caches[i].data_ptr();
    vp[i] = value_caches[i].data_ptr();
  }
  Tensor bm = block_mapping.contiguous();
  TORCH_CHECK(bm.scalar_type() == at::kLong && bm.is_cuda());
  Guard g(key_caches[0]);
  const int64_t bytes = key_caches[0][0].numel() * key_caches[0].element_size();
  ok(mi355x_copy_blocks(kp.data(), vp.data(), num_layers, bm.data_ptr<int64_t>(), bm.size(0),
                        bytes, stream_of(key_caches[0])),
     "copy_blocks");
}

void reshape_and_cache(Tensor& key, Tensor& value, Tensor& key_cache, Tensor& value_cache,
                       Tensor& slot_mapping, const std::string& kv_cache_dtype, Tensor& k_scale,
                       Tensor& v_scale) {
  const int kvd = kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale);
  Guard g(key);
  ok(mi355x_reshape_and_cache(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
                              value_cache.data_ptr(), slot_mapping.data_ptr<int64_t>(),
                              slot_mapping.size(0), key.stride(0), value.stride(0), key.size(1),
                              key.size(2), key_cache.size(3), key_cache.size(4), dt(key), kvd,
                              scale_ptr(kvd, k_scale), scale_ptr(kvd, v_scale), stream_of(key)),
     "reshape_and_cache");
}

void reshape_and_cache_flash(Tensor& key, Tensor& value, Tensor& key_cache, Tensor& value_cache,
                             Tensor& slot_mapping, const std::string& kv_cache_dtype,
                             Tensor& k_scale, Tensor& v_scale) {
  const int kvd = kv_dtype(kv_cache_dtype, key_cache, k_scale, v_scale);
  TORCH_CHECK(key_cache.stride(0) == value_cache.stride(0));
  Guard g(key);
  ok(mi355x_reshape_and_cache_flash(
         key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
         slot_mapping.data_ptr<int64_t>(), slot_mapping.size(0), key_cache.stride(0),
         key_cache.stride(1), key_cache.stride(2), key.stride(0), value.stride(0), key.size(1),
         key.size(2), key_cache.size(1), dt(key), kvd, scale_ptr(kvd, k_scale),
         scale_ptr(kvd, v_scale), stream_of(key)),
     "reshape_and_cache_flash");
}

// ref: csrc/cache_kernels.cu:564-612 ("Only for testing" there)
void convert_fp8(Tensor& dst_cache, Tensor& src_cache, double scale, const std::string& kv_cache_dtype) {
  TORCH_CHECK(src_cache.is_cuda(), "src must be on a GPU");
  TORCH_CHECK(dst_cache.is_cuda(), "dst must be on a GPU");
  TORCH_CHECK(src_cache.get_device() == dst_cache.get_device(), "src and dst must be on the same GPU");
  TORCH_CHECK(kv_cache_dtype == "fp8" || kv_cache_dtype == "fp8_e4m3" || kv_cache_dtype == "fp8_e5m2",
              "Unsupported data type: ", kv_cache_dtype);
  TORCH_CHECK(src_cache.numel() == dst_cache.numel() && src_cache.is_contiguous() &&
                  dst_cache.is_contiguous(),
              "convert_fp8: contiguous tensors of equal size expected");
  const bool to_fp8 = dst_cache.element_size() == 1;
  TORCH_CHECK((to_fp8 ? src_cache : dst_cache).element_size() != 1 &&
                  (to_fp8 ? dst_cache : src_cache).element_size() == 1,
              "convert_fp8: exactly one side must be a 1-byte (fp8) tensor");
  Guard g(src_cache);
  ok(mi355x_convert_fp8(dst_cache.data_ptr(), src_cache.data_ptr(), src_cache.numel(), (float)scale,
                        (to_fp8 ? 1 : 0) + (kv_cache_dtype == "fp8_e5m2" ? 2 : 0),
                        dt(to_fp8 ? src_cache : dst_cache), stream_of(src_cache)),
     "convert_fp8");
}

// ------------------------------------------------------------------ cuda utils
int64_t get_device_attribute(int64_t attribute, int64_t device_id) {
  const int64_t v = mi355x_get_device_attribute(attribute, device_id);
  TORCH_CHECK(v >= 0, mi355x_last_error());
  return v;
}

int64_t get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  const int64_t v = mi355x_get_max_shared_memory_per_block_device_attribute(device_id);
  TORCH_CHECK(v >= 0, mi355x_last_error());
  return v;
}

// weak_ref_tensor: a non-owning alias (used by vLLM's graph capture); schema :35-36.
Tensor weak_ref_tensor(Tensor& tensor) {
  TORCH_CHECK(tensor.is_cuda(), "weak_ref_tensor expects a GPU tensor");
  return at::from_blob(tensor.data_ptr(), tensor.sizes(), tensor.strides(), tensor.options());
}

}  // namespace

TORCH_LIBRARY(_C, ops) {
  ops.def("weak_ref_tensor(Tensor input) -> Tensor");
  ops.impl("weak_ref_tensor", c10::kCUDA, &weak_ref_tensor);

  ops.def(
      "paged_attention_v1("
      "    Tensor! out, Tensor query, Tensor key_cache,"
      "    Tensor value_cache, int num_kv_heads, float scale,"
      "    Tensor block_tables, Tensor seq_lens, int block_size,"
      "    int max_seq_len, Tensor? alibi_slopes,"
      "    str kv_cache_dtype, Tensor k_scale, Tensor v_scale,"
      "    int tp_rank, int blocksparse_local_blocks,"
      "    int blocksparse_vert_stride, int blocksparse_block_size,"
      "    int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v1", c10::kCUDA, &paged_attention_v1);

  ops.def(
      "paged_attention_v2("
      "    Tensor! out, Tensor! exp_sums, Tensor! max_logits,"
      "    Tensor! tmp_out, Tensor query, Tensor key_cache,"
      "    Tensor value_cache, int num_kv_heads, float scale,"
      "    Tensor block_tables, Tensor seq_lens, int block_size,"
      "    int max_seq_len, Tensor? alibi_slopes,"
      "    str kv_cache_dtype, Tensor k_scale, Tensor v_scale,"
      "    int tp_rank, int blocksparse_local_blocks,"
      "    int blocksparse_vert_stride, int blocksparse_block_size,"
      "    int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v2", c10::kCUDA, &paged_attention_v2);

  ops.def(
      "merge_attn_states("
      "    Tensor! output,"
      "    Tensor!? output_lse,"
      "    Tensor prefix_output,"
      "    Tensor prefix_lse,"
      "    Tensor suffix_output,"
      "    Tensor suffix_lse) -> ()");
  ops.impl("merge_attn_states", c10::kCUDA, &merge_attn_states);

  ops.def("silu_and_mul(Tensor! result, Tensor input) -> ()");
  ops.impl("silu_and_mul", c10::kCUDA, &silu_and_mul);
  ops.def("silu_and_mul_quant(Tensor! result, Tensor input, Tensor scale) -> ()");
  ops.impl("silu_and_mul_quant", c10::kCUDA, &silu_and_mul_quant);

  ops.def("rms_norm(Tensor! result, Tensor input, Tensor weight, float epsilon) -> ()");
  ops.impl("rms_norm", c10::kCUDA, &rms_norm);

  ops.def(
      "fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, "
      "float epsilon) -> ()");
  ops.impl("fused_add_rms_norm", c10::kCUDA, &fused_add_rms_norm);

  ops.def(
      "rms_norm_static_fp8_quant(Tensor! result, Tensor input, Tensor weight, "
      "Tensor scale, float epsilon) -> ()");
  ops.impl("rms_norm_static_fp8_quant", c10::kCUDA, &rms_norm_static_fp8_quant);

  ops.def(
      "fused_add_rms_norm_static_fp8_quant(Tensor! result, Tensor input, "
      "Tensor! residual, Tensor weight, Tensor scale, float epsilon) -> ()");
  ops.impl("fused_add_rms_norm_static_fp8_quant", c10::kCUDA,
           &fused_add_rms_norm_static_fp8_quant);

  ops.def(
      "rms_norm_dynamic_per_token_quant(Tensor! result, Tensor input, "
      "Tensor weight, Tensor! scale, float epsilon, "
      "Tensor? scale_ub, Tensor!? residual) -> ()");
  ops.impl("rms_norm_dynamic_per_token_quant", c10::kCUDA, &rms_norm_dynamic_per_token_quant);

  ops.def(
      "rotary_embedding(Tensor positions, Tensor! query,"
      "                 Tensor!? key, int head_size,"
      "                 Tensor cos_sin_cache, bool is_neox) -> ()");
  ops.impl("rotary_embedding", c10::kCUDA, &rotary_embedding);

  ops.def(
      "batched_rotary_embedding(Tensor positions, Tensor! query,"
      "                         Tensor!? key, int head_size,"
      "                         Tensor cos_sin_cache, bool is_neox,"
      "                         int rot_dim,"
      "                         Tensor cos_sin_cache_offsets) -> ()");
  ops.impl("batched_rotary_embedding", c10::kCUDA, &batched_rotary_embedding);

  ops.def(
      "awq_gemm(Tensor _in_feats, Tensor _kernel, Tensor _scaling_factors, "
      "Tensor _zeros, SymInt split_k_iters, Tensor _temp_space, bool "
      "dtype_bf16) -> Tensor");
  ops.impl("awq_gemm", c10::kCUDA, &awq_gemm);

  ops.def(
      "awq_dequantize(Tensor _kernel, Tensor _scaling_factors, "
      "Tensor _zeros, SymInt split_k_iters, int thx, int thy) -> Tensor");
  ops.impl("awq_dequantize", c10::kCUDA, &awq_dequantize);

  ops.def("awq_to_gptq_4bit(Tensor qweight) -> Tensor");
  ops.impl("awq_to_gptq_4bit", c10::kCUDA, &awq_to_gptq_4bit);

  ops.def(
      "static_scaled_int8_quant(Tensor! result, Tensor input, Tensor scale,"
      "Tensor? azp) -> ()");
  ops.impl("static_scaled_int8_quant", c10::kCUDA, &static_scaled_int8_quant);
  ops.def(
      "dynamic_scaled_int8_quant(Tensor! result, Tensor input, Tensor! scale, "
      "Tensor!? azp) -> ()");
  ops.impl("dynamic_scaled_int8_quant", c10::kCUDA, &dynamic_scaled_int8_quant);

  ops.def(
      "cutlass_scaled_mm(Tensor! out, Tensor a,"
      "                  Tensor b, Tensor a_scales,"
      "                  Tensor b_scales, Tensor? bias) -> ()");
  ops.impl("cutlass_scaled_mm", c10::kCUDA, &cutlass_scaled_mm);

  ops.def("cutlass_scaled_mm_supports_fp8(int cuda_device_capability) -> bool");
  ops.impl("cutlass_scaled_mm_supports_fp8", &cutlass_scaled_mm_supports_fp8);

  ops.def(
      "gptq_gemm(Tensor a, Tensor b_q_weight, Tensor b_gptq_qzeros, "
      "Tensor b_gptq_scales, Tensor b_g_idx, bool use_exllama, int bit, int "
      "group_size, Tensor perm_space, "
      "Tensor temp_space, bool dtype_bf16)-> Tensor");
  ops.impl("gptq_gemm", c10::kCUDA, &gptq_gemm);

  ops.def("gptq_shuffle(Tensor! q_weight, Tensor q_perm, int bit) -> ()");
  ops.impl("gptq_shuffle", c10::kCUDA, &gptq_shuffle);

  ops.def("static_scaled_fp8_quant(Tensor! result, Tensor input, Tensor scale) -> ()");
  ops.impl("static_scaled_fp8_quant", c10::kCUDA, &static_scaled_fp8_quant);

  ops.def("dynamic_scaled_fp8_quant(Tensor! result, Tensor input, Tensor! scale) -> ()");
  ops.impl("dynamic_scaled_fp8_quant", c10::kCUDA, &dynamic_scaled_fp8_quant);

  ops.def(
      "dynamic_per_token_scaled_fp8_quant(Tensor! result, Tensor input, "
      "Tensor! scale, Tensor? scale_ub) -> ()");
  ops.impl("dynamic_per_token_scaled_fp8_quant", c10::kCUDA,
           &dynamic_per_token_scaled_fp8_quant);
}

TORCH_LIBRARY(_C_cache_ops, cache_ops) {
  cache_ops.def("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()");
  cache_ops.impl("swap_blocks", c10::kCUDA, &swap_blocks);

  cache_ops.def(
      "copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches, "
      "Tensor block_mapping) -> ()");
  cache_ops.impl("copy_blocks", c10::kCUDA, &copy_blocks);

  cache_ops.def(
      "reshape_and_cache(Tensor key, Tensor value,"
      "                  Tensor! key_cache, Tensor! value_cache,"
      "                  Tensor slot_mapping,"
      "                  str kv_cache_dtype,"
      "                  Tensor k_scale, Tensor v_scale) -> ()");
  cache_ops.impl("reshape_and_cache", c10::kCUDA, &reshape_and_cache);

  cache_ops.def(
      "reshape_and_cache_flash(Tensor key, Tensor value,"
      "                        Tensor! key_cache,"
      "                        Tensor! value_cache,"
      "                        Tensor slot_mapping,"
      "                        str kv_cache_dtype,"
      "                        Tensor k_scale, Tensor v_scale) -> ()");
  cache_ops.impl("reshape_and_cache_flash", c10::kCUDA, &reshape_and_cache_flash);

  cache_ops.def(
      "convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, "
      "str kv_cache_dtype) -> ()");
  cache_ops.impl("convert_fp8", c10::kCUDA, &convert_fp8);
}

TORCH_LIBRARY(_C_cuda_utils, cuda_utils) {
  cuda_utils.def("get_device_attribute(int attribute, int device_id) -> int");
  cuda_utils.impl("get_device_attribute", &get_device_attribute);

  cuda_utils.def("get_max_shared_memory_per_block_device_attribute(int device_id) -> int");
  cuda_utils.impl("get_max_shared_memory_per_block_device_attribute",
                  &get_max_shared_memory_per_block_device_attribute);
}

// `import vllm_metax_amd._C` loads this library (ref: csrc/core/registration.h:22-27).
PyMODINIT_FUNC PyInit__C() {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_C", nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
