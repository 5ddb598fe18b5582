/*
 * mi355x_hotpath.h — C-ABI of the MI355X (gfx950) quantized-inference hot path.
 *
 * This is the drop-in boundary: every entry point takes plain device pointers,
 * sizes, strides (in ELEMENTS unless the name says bytes) and a HIP stream, and
 * corresponds to one launcher of the reference extension (vLLM-metax `_C.so`).
 * The citation after each declaration names the reference launcher whose
 * pointer/stride extraction fixes the parameter list (paths relative to
 * /root/reference).
 *
 * Conventions
 *   - return 0 on success, a negative MI355X_E* code otherwise;
 *     mi355x_last_error() returns a thread-local description of the last error.
 *   - no allocation, no host sync, no global state; every launch goes to `stream`
 *     (hipStream_t passed as void*), so every call is hipGraph-capturable unless
 *     the comment says otherwise.
 *   - dtype is passed as mi355x_dtype (no templates cross the ABI).
 *   - "Tensor!" arguments of the reference schema are the non-const pointers.
 */
#ifndef MI355X_HOTPATH_H_
#define MI355X_HOTPATH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_ABI_VERSION 5   /* 5: + paged_attention_v2_ps, scaled_mm_fp8_deferred and its slab consumers, scaled_mm_prepack(ed), scaled_mm_split_elems, w4a16_prepacked_split_elems (additive; w4a16_gemm_prepacked takes m >= 384); 4: + paged_prefill_attention_alibi, silu_and_mul_per_token_quant (additive); 3: + greedy_advance, paged_attention_fused_qkv, *_rms_norm_image, paged_prefill_attention_image (additive);
                                * 2 was BREAKING (kv_cache_dtype / k_scale / v_scale inserted before `stream` in reshape_and_cache*,
                                * paged_attention_v1/_v2, paged_prefill_attention): a binding must refuse a library whose
                                * mi355x_abi_version() differs from the version it was written for (vllm_metax_amd/_abi.py does) */

typedef enum {
  MI355X_F16 = 0,  /* IEEE half  (torch.float16)  */
  MI355X_BF16 = 1, /* bfloat16   (torch.bfloat16) */
  MI355X_F32 = 2   /* float      (torch.float32)  */
} mi355x_dtype;

enum {
  MI355X_OK = 0,
  MI355X_EINVAL = -1,      /* bad argument (null pointer, negative size, misalignment) */
  MI355X_EUNSUPPORTED = -2,/* shape / dtype / feature outside the supported set        */
  MI355X_ELAUNCH = -3      /* hipLaunch / hipMemcpyAsync reported an error             */
};

/* kv_cache_dtype of the cache / attention entry points (the schema's `str kv_cache_dtype`):
 * "auto" = the cache holds scalar_t; "fp8" / "fp8_e4m3" = OCP e4m3fn bytes, x = 16
 * (cache byte = sat_e4m3(float(x) / *k_scale), value = float(byte) * *k_scale; likewise V).
 * The reference's dispatch accepts only "auto" (csrc/quantization/fp8/metax/quant_utils.cuh:29-42);
 * the fp8 cache is SURVEY §8f-3, hook points csrc/cache_kernels.cu:245-253,258-269. */
typedef enum {
  MI355X_KV_AUTO = 0,
  MI355X_KV_FP8_E4M3 = 1,
  MI355X_KV_FP8_E5M2 = 2   /* "fp8_e5m2": e5m2 bytes, same layouts and scales as e4m3 (round 3) */
} mi355x_kv_cache_dtype;

typedef void* mi355x_stream; /* hipStream_t */

int mi355x_abi_version(void);
const char* mi355x_last_error(void);

/* ------------------------------------------------------------------ utils --
 * ref: csrc/cuda_utils_kernels.cu (get_device_attribute,
 *      get_max_shared_memory_per_block_device_attribute), schema
 *      csrc/torch_bindings.cpp:461-468. Returns the value, or a negative code. */
int64_t mi355x_get_device_attribute(int64_t attribute, int64_t device_id);
int64_t mi355x_get_max_shared_memory_per_block_device_attribute(int64_t device_id);

/* ------------------------------------------------------------- KV cache ops --
 * reshape_and_cache: scatter T new K/V rows into the paged "x-split" layout
 *   key_cache   [num_blocks, num_heads, head_size/x, block_size, x]
 *   value_cache [num_blocks, num_heads, head_size, block_size]
 * slot < 0 => token skipped. Pure copy (bit-exact).
 * ref: csrc/cache_kernels.cu:407-433 (launcher), :203-255 (kernel). */
int mi355x_reshape_and_cache(const void* key, const void* value, void* key_cache,
                             void* value_cache, const int64_t* slot_mapping,
                             int num_tokens, int64_t key_stride, int64_t value_stride,
                             int num_heads, int head_size, int block_size, int x,
                             int dtype, int kv_cache_dtype, const float* k_scale,
                             const float* v_scale, mi355x_stream stream);

/* reshape_and_cache_flash: same into NHD [num_blocks, block_size, heads, head_size]
 * (or HND through the strides). num_tokens = slot_mapping.size(0).
 * ref: csrc/cache_kernels.cu:450-488 (launcher), :271-344 (kernel). */
int mi355x_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                   void* value_cache, const int64_t* slot_mapping,
                                   int num_tokens, int64_t block_stride,
                                   int64_t page_stride, int64_t head_stride,
                                   int64_t key_stride, int64_t value_stride,
                                   int num_heads, int head_size, int block_size,
                                   int dtype, int kv_cache_dtype, const float* k_scale,
                                   const float* v_scale, mi355x_stream stream);

/* convert_fp8: elementwise over a flat cache of `numel` elements.  to_fp8 = 1: src scalar_t ->
 * dst e4m3 byte = sat(float(x) / scale); 0: src e4m3 byte -> dst scalar_t = T(float(byte) * scale); 3 / 2: the same
 * two directions for e5m2 bytes ("fp8_e5m2").
 * ref: csrc/cache_kernels.cu:544-612 ("only for testing" there), schema torch_bindings.cpp:424. */
int mi355x_convert_fp8(void* dst, const void* src, int64_t numel, float scale, int to_fp8,
                       int dtype, mi355x_stream stream);

/* copy_blocks: for every layer l and pair p copy block src->dst inside
 * key_caches[l] and value_caches[l]. `key_cache_ptrs`/`value_cache_ptrs` are HOST
 * arrays of device pointers (passed to the kernel by value, so unlike the
 * reference there is no H2D upload and no sync); `block_mapping` is a DEVICE
 * int64 [num_pairs, 2]. bytes_per_block = numel_per_block * element_size.
 * ref: csrc/cache_kernels.cu:116-163 (launcher), :65-91 (kernel). */
int mi355x_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs,
                       int num_layers, const int64_t* block_mapping, int num_pairs,
                       int64_t bytes_per_block, mi355x_stream stream);

/* swap_blocks: block copies between two caches; `block_mapping` is a HOST int64
 * [num_pairs, 2]; kind: 0 = D2D, 1 = D2H, 2 = H2D. Runs of consecutive pairs are
 * merged into one hipMemcpyAsync. Not graph-capturable when host memory is pageable.
 * ref: csrc/cache_kernels.cu:18-60. */
int mi355x_swap_blocks(const void* src, void* dst, const int64_t* block_mapping,
                       int num_pairs, int64_t block_size_in_bytes, int kind,
                       mi355x_stream stream);

/* --------------------------------------------------------- paged attention --
 * Single-query (decode) attention over the x-split paged KV cache.
 *   out/query [num_seqs, num_heads, head_size] (query row stride = q_stride)
 *   block_tables int32 [num_seqs, max_num_blocks_per_seq]; seq_lens int32 [num_seqs]
 *   alibi_slopes float [num_heads] or NULL. kv_cache_dtype: mi355x_kv_cache_dtype; k_scale /
 *   v_scale: device pointers to one float each (read only for the fp8 cache; block sizes 16, 32).
 * One workgroup serves ALL query heads of one KV head (GQA reuse), unlike the
 * reference's one block per query head.
 * ref: csrc/attention/paged_attention_v1.cu:43-125,160-182;
 *      csrc/attention/attention_kernels.cuh:75-485. */
int mi355x_paged_attention_v1(void* out, const void* query, const void* key_cache,
                              const void* value_cache, int num_seqs, int num_heads,
                              int num_kv_heads, int head_size, int block_size,
                              float scale, const int* block_tables,
                              const int* seq_lens, int max_num_blocks_per_seq,
                              int max_seq_len, const float* alibi_slopes,
                              int64_t q_stride, int64_t kv_block_stride,
                              int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                              const float* k_scale, const float* v_scale, mi355x_stream stream);

/* Largest max_seq_len mi355x_paged_attention_v1 accepts for this head geometry: v1 keeps the
 * logits of a whole sequence for all (<= 4) query heads of a KV head in one workgroup's LDS
 * (160 KiB); beyond it the caller must use _v2.  The reference sizes its LDS the same way
 * (csrc/attention/paged_attention_v1.cu:77-87) and leaves the v1 / v2 choice to its caller.
 * Pure host function (no launch).  Returns the limit (a multiple of 64) or a negative code. */
int mi355x_paged_attention_v1_max_seq_len(int num_seqs, int num_heads, int num_kv_heads,
                                          int head_size, int block_size, int dtype);

/* Split-KV variant: partitions of MI355X_PA_PARTITION_SIZE tokens, then an
 * LSE-rescaled reduce.  exp_sums/max_logits float [num_seqs, num_heads, P],
 * tmp_out [num_seqs, num_heads, P, head_size], P = ceil(max_seq_len / 512).
 * ref: csrc/attention/paged_attention_v2.cu:43-131,167-192;
 *      csrc/attention/attention_kernels.cuh:519-658. */
#define MI355X_PA_PARTITION_SIZE 512
int mi355x_paged_attention_v2(void* out, float* exp_sums, float* max_logits,
                              void* tmp_out, const void* query, const void* key_cache,
                              const void* value_cache, int num_seqs, int num_heads,
                              int num_kv_heads, int head_size, int block_size,
                              float scale, const int* block_tables,
                              const int* seq_lens, int max_num_blocks_per_seq,
                              int max_seq_len, const float* alibi_slopes,
                              int64_t q_stride, int64_t kv_block_stride,
                              int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                              const float* k_scale, const float* v_scale, mi355x_stream stream);
/* The same with the partition size chosen by the caller (a positive multiple of the block size and of 16;
 * exp_sums / max_logits / tmp_out sized for P = ceil(max_seq_len / partition_size)).  The reference fixes 512
 * (paged_attention_v2.cu:45); with few (sequence, kv head) pairs — one TP = 8 rank of a 70B model at batch 64
 * has 64 — 512-token partitions leave most CUs without a workgroup, so the host side
 * (attention/backend.py::decode_partition_size) splits finer.  mi355x_paged_attention_v2 == partition_size 512. */
int mi355x_paged_attention_v2_ps(void* out, float* exp_sums, float* max_logits,
                                 void* tmp_out, const void* query, const void* key_cache,
                                 const void* value_cache, int num_seqs, int num_heads,
                                 int num_kv_heads, int head_size, int block_size,
                                 float scale, const int* block_tables,
                                 const int* seq_lens, int max_num_blocks_per_seq,
                                 int max_seq_len, const float* alibi_slopes,
                                 int64_t q_stride, int64_t kv_block_stride,
                                 int64_t kv_head_stride, int dtype, int kv_cache_dtype,
                                 const float* k_scale, const float* v_scale, int partition_size,
                                 mi355x_stream stream);

/* Paged prefill / chunked prefill (varlen, causal bottom-right aligned, GQA) over
 * the SAME x-split paged cache (the new tokens were already written by
 * reshape_and_cache).  q/out [total_q, num_heads, head_size] packed by
 * cu_seqlens_q int32 [num_seqs+1]; seq_lens int32 [num_seqs] = context + new.
 * sliding_window W > 0: a query sees only its last W keys (the call site's window_size = (W-1, 0));
 * softcap c > 0: scores become c * tanh(s / c) before the mask; 0 = off.  For head_size 128 / block_size 16 /
 * 16-bit types both run on the MFMA kernel (a separate instantiation), other shapes on the general kernel.
 * ref (call site, third-party kernel): vllm_metax/v1/attention/backends/
 *      flash_attn.py:725-747; semantic oracle tests/kernels/attention/
 *      test_flash_attn.py:27-80. */
int mi355x_paged_prefill_attention(void* out, const void* query, const void* key_cache,
                                   const void* value_cache, int num_seqs, int num_heads,
                                   int num_kv_heads, int head_size, int block_size,
                                   float scale, const int* block_tables,
                                   const int* seq_lens, const int* cu_seqlens_q,
                                   int max_query_len, int max_num_blocks_per_seq,
                                   int64_t q_stride, int64_t out_stride,
                                   int64_t kv_block_stride, int64_t kv_head_stride,
                                   int dtype, int kv_cache_dtype, const float* k_scale,
                                   const float* v_scale, int sliding_window, float softcap,
                                   mi355x_stream stream);
/* The same with the call site's alibi_slopes (float [num_heads] or NULL): slope[head] * (key position - query
 * position) is added to the scaled (and capped) score — the bias of the decode kernel
 * (csrc/attention/attention_kernels.cuh:286) at every query position.  MFMA kernel shapes only (ABI 4). */
int mi355x_paged_prefill_attention_alibi(void* out, const void* query, const void* key_cache,
                                         const void* value_cache, int num_seqs, int num_heads,
                                         int num_kv_heads, int head_size, int block_size,
                                         float scale, const int* block_tables,
                                         const int* seq_lens, const int* cu_seqlens_q,
                                         int max_query_len, int max_num_blocks_per_seq,
                                         int64_t q_stride, int64_t out_stride,
                                         int64_t kv_block_stride, int64_t kv_head_stride,
                                         int dtype, int kv_cache_dtype, const float* k_scale,
                                         const float* v_scale, int sliding_window, float softcap,
                                         const float* alibi_slopes, mi355x_stream stream);

/* ------------------------------------------------------------- layernorm --
 * ref: csrc/layernorm_kernels.cu:141-162 (rms_norm), :174-217 (fused_add). */
int mi355x_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                    int num_tokens, int hidden_size, int64_t input_stride, int dtype,
                    mi355x_stream stream);
int mi355x_fused_add_rms_norm(void* input, void* residual, const void* weight,
                              float epsilon, int num_tokens, int hidden_size,
                              int64_t input_stride, int dtype, mi355x_stream stream);

/* MI355X-side fusion (no reference op): fused_add_rms_norm whose input row is the still
 * unreduced output of the preceding decode GEMM, x = T(slab[0] + .. + slab[sk-1]) with fp32 slabs
 * [sk][num_tokens][hidden] as left by mi355x_awq_gemm_deferred.
 * Bit-identical to gemm + fused_add_rms_norm (same summation order, same rounding points); saves
 * the slab-sum launch.  sk == 0: plain fused_add_rms_norm on `input`.  `input` receives the
 * normalised rows either way.  ref semantics: csrc/layernorm_kernels.cu:54-138. */
int mi355x_fused_add_rms_norm_slabs(void* input, void* residual, const void* weight,
                                    const float* slabs, int sk, float epsilon, int num_tokens,
                                    int hidden_size, int64_t input_stride, int dtype,
                                    mi355x_stream stream);

/* ref: csrc/layernorm_quant_kernels.cu:168-194, :210-249. out is float8_e4m3fn. */
int mi355x_rms_norm_static_fp8_quant(void* out, const void* input, const void* weight,
                                     const float* scale, float epsilon, int num_tokens,
                                     int hidden_size, int64_t input_stride, int dtype,
                                     mi355x_stream stream);
int mi355x_fused_add_rms_norm_static_fp8_quant(void* out, void* input, void* residual,
                                               const void* weight, const float* scale,
                                               float epsilon, int num_tokens,
                                               int hidden_size, int64_t input_stride,
                                               int dtype, mi355x_stream stream);
/* ref: csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu
 *      :132-155; helpers layernorm_utils.cuh:17-115. residual/scale_ub may be NULL. */
int mi355x_rms_norm_dynamic_per_token_quant(void* out, const void* input,
                                            const void* weight, float* scales,
                                            float epsilon, const float* scale_ub,
                                            void* residual, int num_tokens,
                                            int hidden_size, int dtype,
                                            mi355x_stream stream);

/* ---------------------------------------------------------------- fp8 quant --
 * ref: csrc/quantization/fp8/common.cu:137-168, :170-212, :214-248. Strides in
 * elements. dynamic_scaled: `scale` must be zero-initialised by the caller (the
 * kernel folds absmax/448 in with an atomic max, as the reference does). */
int mi355x_static_scaled_fp8_quant(void* out, const void* input, const float* scale,
                                   int num_tokens, int hidden_size,
                                   int64_t in_row_stride, int64_t out_row_stride,
                                   int dtype, mi355x_stream stream);
int mi355x_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale,
                                    int num_tokens, int hidden_size,
                                    int64_t in_row_stride, int64_t out_row_stride,
                                    int dtype, mi355x_stream stream);
int mi355x_dynamic_per_token_scaled_fp8_quant(void* out, const void* input,
                                              float* scales, const float* scale_ub,
                                              int num_tokens, int hidden_size,
                                              int64_t in_row_stride,
                                              int64_t out_row_stride, int dtype,
                                              mi355x_stream stream);

/* ------------------------------------------------------------------ rotary --
 * In-place RoPE on query and (optional) key. positions int64 [num_tokens].
 * ref: csrc/pos_encoding_kernels.cu:133-213 (launcher), :10-100 (math). */
int mi355x_rotary_embedding(const int64_t* positions, void* query, void* key,
                            const void* cos_sin_cache, int num_tokens, int rot_dim,
                            int64_t query_stride, int64_t key_stride,
                            int64_t head_stride, int num_heads, int num_kv_heads,
                            int head_size, int is_neox, int dtype,
                            mi355x_stream stream);
/* batched_rotary_embedding (several LoRA rope tables stacked in one cache): row = positions[t] +
 * cos_sin_cache_offsets[t]; rot_dim is passed explicitly as in the reference schema.
 * ref: csrc/pos_encoding_kernels.cu:102-129 (kernel), :219-306 (launcher), schema torch_bindings.cpp:224-229. */
int mi355x_batched_rotary_embedding(const int64_t* positions, void* query, void* key,
                                    const void* cos_sin_cache,
                                    const int64_t* cos_sin_cache_offsets, int num_tokens,
                                    int rot_dim, int64_t query_stride, int64_t key_stride,
                                    int64_t head_stride, int num_heads, int num_kv_heads,
                                    int head_size, int is_neox, int dtype, mi355x_stream stream);


/* -------------------------------------------------------------- activation --
 * out[t, :d] = silu(in[t, :d]) * in[t, d:2d].
 * ref: csrc/activation_kernels.cu:24-36,142-147,243-247. */
int mi355x_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype,
                        mi355x_stream stream);

/* out (fp8 e4m3, [tokens, d]) = fp8_sat(T(silu(x) * y) * (1 / *scale)): the gated activation with
 * the static per-tensor fp8 quantisation of the following FP8 GEMM fused in. input fp16 / bf16.
 * ref: csrc/quantization/activation_kernels.cu:21-90 (kernel), :117-127 (host),
 * schema csrc/torch_bindings.cpp:115-117. */
int mi355x_silu_and_mul_quant(void* out, const void* input, const float* scale, int num_tokens,
                              int d, int dtype, mi355x_stream stream);

/* ------------------------------------------------------- int4 weight-only --
 * awq_to_gptq_4bit: AWQ qweight [K, N/8] (N-interleaved nibbles) -> exllama
 * layout, memory [K/8, N] words, nibble p of word (kk,n) = W[8kk + {0,2,4,6,1,3,5,7}[p], n].
 * ref: csrc/quantization/awq/gemm_kernels.cu:127-184 (kernel), :323-356. */
int mi355x_awq_to_gptq_4bit(uint32_t* out, const uint32_t* qweight, int k, int n,
                            mi355x_stream stream);

/* awq_dequantize: original AWQ layout -> dense [K, N] of `dtype`.
 * ref: csrc/quantization/awq/gemm_kernels.cu:358-402 (launcher), :96-125. */
int mi355x_awq_dequantize(void* out, const uint32_t* qweight, const void* scales,
                          const uint32_t* qzeros, int k, int n, int group_size,
                          int dtype, mi355x_stream stream);

/* awq_gemm: C[M,N] = A[M,K] . ((Q - Z) * S); Q in the exllama layout produced by
 * awq_to_gptq_4bit, qzeros [K/g, N/8] in AWQ nibble order, scales [K/g, N].
 * `workspace` (float, may be NULL: then K is never split across workgroups) is the
 * reference's temp_space (awq.py:140-147).  For M <= 64 the kernel may split K over up to
 * workspace_elems / (min(m,64)*n) workgroups, each writing one fp32 partial slab [m, n] into it;
 * a second kernel adds the slabs in a fixed order and rounds once (no atomics: results do not
 * depend on timing).  8*m*n elements allow every split the planner wants.  It need not be
 * zeroed; contents on exit are unspecified.
 * `dq_workspace` (16-byte aligned, may be NULL) is scratch for re-tiled operands:
 *   - M <= 64: >= roundup(m,16)*k*2 bytes: the activations are re-tiled into MFMA operand
 *     images, which selects the LDS-DMA "stripe" decode kernel (k % 128 == 0, group 32/64/128k);
 *   - M >= 1024: >= (n + roundup(m,16))*k*2 bytes: weights are dequantised once and
 *     activations re-tiled, then a pure-MFMA GEMM follows (same numerics).
 *   Without it the register-staged kernels are used.  Contents on exit are unspecified.
 * ref: csrc/quantization/awq/gemm_kernels.cu:410-463, :283-318, :186-281. */
int mi355x_awq_gemm(void* c, const void* a, const uint32_t* qweight, const void* scales,
                    const uint32_t* qzeros, float* workspace, int64_t workspace_elems,
                    void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k,
                    int group_size, int64_t lda, int dtype, mi355x_stream stream);

/* awq_gemm whose split-K reduction is left to the consumer: identical arguments plus `sk_out`.
 * If the kernel split K (M <= 64, workspace given) the fp32 partial slabs [sk][m][n] stay in
 * `workspace`, `c` is NOT written and *sk_out = sk (>= 2); otherwise `c` holds the result and
 * *sk_out = 0.  Consumer: mi355x_fused_add_rms_norm_slabs. */
int mi355x_awq_gemm_deferred(void* c, const void* a, const uint32_t* qweight, const void* scales,
                             const uint32_t* qzeros, float* workspace, int64_t workspace_elems,
                             void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k,
                             int group_size, int64_t lda, int dtype, int* sk_out,
                             mi355x_stream stream);

/* MI355X-side prefill fusion: out[M, N/2] = silu_and_mul(awq_gemm(a, W_gate_up)) in one GEMM launch
 * (the epilogue applies csrc/activation_kernels.cu:14-36 to the T-rounded accumulators: bit-identical
 * to awq_gemm followed by silu_and_mul, without writing and re-reading the [M, N] intermediate).
 * m <= 64 (decode: n % 128 == 0, k % 128 == 0, group 32/64/128k; dq_workspace unused) or
 * m >= 1024 (prefill: n % 256 == 0, dq_workspace >= (n + roundup(m,16))*k*2 bytes);
 * otherwise MI355X_EUNSUPPORTED (call the two ops). */
int mi355x_awq_gemm_silu_mul(void* out, const void* a, const uint32_t* qweight, const void* scales,
                             const uint32_t* qzeros, void* dq_workspace, int64_t dq_workspace_bytes,
                             int m, int n, int k, int group_size, int64_t lda, int dtype,
                             mi355x_stream stream);

/* The same fusion for prefill with the re-tiling of the NEXT GEMM's activations folded in
 * (m >= 1024 only): out_packed receives act = silu_and_mul(awq_gemm(a, W_gate_up)) [m, n/2] not
 * row-major but as the MFMA operand image the prefill GEMM reads its activations from
 * (roundup(m,16) * n/2 elements: [row tile of 16][k tile of 32][64 slots of 16 B], slot
 * 16 lr + (lc ^ {0,12,2,14}[lr]) = act[16 mt + lc][32 kt + 8 lr .. + 7], rows >= m zero), and
 * mi355x_awq_gemm_packed_a consumes such an image as its `a` (then k = n/2 of the producer).
 * Together they are bit-identical to awq_gemm(silu_and_mul(awq_gemm(a, W_gate_up)), W_down)
 * (ref call sites: vllm_metax/quant_config/awq.py:140-147 twice around
 * csrc/activation_kernels.cu:24-36) with one launch and one read + write of act less. */
int mi355x_awq_gemm_silu_mul_packed(void* out_packed, const void* a, const uint32_t* qweight,
                                    const void* scales, const uint32_t* qzeros, void* dq_workspace,
                                    int64_t dq_workspace_bytes, int m, int n, int k, int group_size,
                                    int64_t lda, int dtype, mi355x_stream stream);
int mi355x_awq_gemm_packed_a(void* out, const void* a_packed, const uint32_t* qweight,
                             const void* scales, const uint32_t* qzeros, void* dq_workspace,
                             int64_t dq_workspace_bytes, int m, int n, int k, int group_size,
                             int dtype, mi355x_stream stream);

/* gptq_shuffle: in-place exllama nibble shuffle of q_weight [K/8, N]; with q_perm
 * (int32 [K]) rows are first made sequential through `scratch` (>= K/8*N words).
 * ref: csrc/quantization/gptq/q_gemm.cu:2415-2423, :2321-2368, qdq_4.cuh:16-29,
 *      q_gemm.cu:2145-2174. bit must be 4 or 8. */
int mi355x_gptq_shuffle(uint32_t* q_weight, const int* q_perm, uint32_t* scratch, int k,
                        int n, int bit, mi355x_stream stream);

/* Weight-load-time preprocessing for the prefill GEMM (MI355X-side, no reference op): the prefill path
 * (m >= 1024) multiplies operand images (1-KiB pieces [16-row tile][K/32][64 slots][8 x T], see
 * DESIGN.md §2) and otherwise re-derives the weights' image from the int4 words on EVERY call.
 * mi355x_w4a16_prepack writes that image once: image[n * k] elements of T = T(fma(q, s, -z s)), the
 * reference's dequantised value (hgemm_gptq.h:487-570) bit for bit, so a GEMM on the image is
 * bit-identical to mi355x_awq_gemm / _gptq_gemm at the same m.  qweight is the exllama layout
 * (awq_to_gptq_4bit / gptq_shuffle output); gptq_zeros != 0: zero = qzeros + 1.
 * Costs n * k * sizeof(T) bytes of HBM per layer (Llama-3-8B: 436 MB per decoder layer). */
int mi355x_w4a16_prepack(void* image, const uint32_t* qweight, const void* scales,
                         const uint32_t* qzeros, int n, int k, int group_size, int gptq_zeros, int dtype,
                         mi355x_stream stream);
/* out[m, n] = a[m, k] . W  with W given as a prepacked image; mode bits:
 *   SILU      out = silu_and_mul(a . W) [m, n/2] (gate_up projection, as mi355x_awq_gemm_silu_mul)
 *   OUT_IMAGE with SILU: write it as the operand image of the next GEMM (as .._silu_mul_packed)
 *   A_IMAGE   `a` already is an operand image (as mi355x_awq_gemm_packed_a)
 * a_workspace: roundup(m,16) * k * sizeof(T) bytes for the activation image (unused with A_IMAGE).
 * m >= 384 (the per-call forms mi355x_awq_gemm / _gptq_gemm switch to the image GEMM at m >= 1024, where deriving
 * the image per call pays; with the image at hand 384 rows do).  Shapes with few 256 x 256 output tiles (chunked-
 * prefill sized m, a TP shard's narrow n) split K when a_workspace has room behind the activation image:
 * mi355x_w4a16_prepacked_split_elems returns the 4-byte elements to add for that (0: not split); fp32 partial
 * tiles, added in split order by one more launch (which also applies the SILU epilogue / writes the OUT_IMAGE).
 * The per-call forms (mi355x_awq_gemm / _gptq_gemm / .._silu_mul / .._silu_mul_packed / .._packed_a at m >= 1024)
 * plan the same split from dq_workspace bytes beyond their operand images, so that all forms agree bit for bit
 * when each is given those elements. */
enum { MI355X_PREPACKED_SILU = 1, MI355X_PREPACKED_OUT_IMAGE = 2, MI355X_PREPACKED_A_IMAGE = 4 };
int64_t mi355x_w4a16_prepacked_split_elems(int m, int n, int k);
int mi355x_w4a16_gemm_prepacked(void* out, const void* a, const void* image, void* a_workspace,
                                int64_t a_workspace_bytes, int m, int n, int k, int64_t lda, int mode,
                                int dtype, mi355x_stream stream);

/* gptq_gemm: C = A[:, perm] . ((Q - (Z + 1)) * S), 4-bit, shuffled layout.
 * g_idx == NULL => no act-order; else g_idx is the argsort permutation and
 * perm_space (>= m*k elements of 2 bytes) receives the permuted activations.
 * ref: csrc/quantization/gptq/q_gemm.cu:2373-2413, :1983-2007, :1770-1786. */
int mi355x_gptq_gemm(void* c, const void* a, const uint32_t* qweight,
                     const uint32_t* qzeros, const void* scales, const int* g_idx,
                     void* perm_space, float* workspace, int64_t workspace_elems,
                     void* dq_workspace, int64_t dq_workspace_bytes, int m, int n, int k, int bit,
                     int group_size, int dtype, mi355x_stream stream);

/* merge_attn_states: out = softmax-weighted combination of two partial
 * attention results over disjoint KV ranges, from their log-sum-exps; lse == +inf counts as -inf.
 * output / prefix_output / suffix_output [num_tokens, num_heads, head_size] (heads contiguous),
 * *_lse [num_heads, num_tokens] fp32, output_lse may be NULL.  head_size %% (16/sizeof(T)) == 0.
 * ref: csrc/attention/merge_attn_states.cu:15-87 (kernel), :133-172 (launcher);
 *      schema csrc/torch_bindings.cpp:74-82. */
int mi355x_merge_attn_states(void* output, float* output_lse, const void* prefix_output,
                             const float* prefix_lse, const void* suffix_output,
                             const float* suffix_lse, int num_tokens, int num_heads, int head_size,
                             int dtype, mi355x_stream stream);

/* silu_and_mul + dynamic per-token fp8 quantisation in one launch (MI355X-side fusion for the input of an fp8
 * down_proj; ABI 4): out fp8 [num_tokens, d], scales float [num_tokens], input [num_tokens, 2 d].  The bits of
 * mi355x_silu_and_mul followed by mi355x_dynamic_per_token_scaled_fp8_quant (no scale_ub).  Returns 1 (no error)
 * when the fused form does not apply (d % 8 != 0, d > 16384, unaligned pointers): run the two ops. */
int mi355x_silu_and_mul_per_token_quant(void* out, float* scales, const void* input, int num_tokens, int d,
                                        int dtype, mi355x_stream stream);

/* ------------------------------------------------------- int8 W8A8 (§8f-4) --
 * scaled_mm_int8: the int8 branch of cutlass_scaled_mm — out[M,N] (bf16/f16) =
 *   a_scales . (a[M,K] int8 row-major x b[K,N] int8 COLUMN-major) . b_scales (+ bias[N]), exact
 *   int32 accumulation, epilogue a_s * (b_s * float(acc)) + bias in fp32, one rounding.
 *   Arguments as mi355x_scaled_mm_fp8 (workspace: sk*m*n 4-byte elements for an sk-way M <= 64 split-K).
 * ref: csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu:34-39, :84-140; schema
 *      csrc/torch_bindings.cpp:251-256.
 * static / dynamic_scaled_int8_quant: q = clamp(rint(x / scale), -127, 127); dynamic computes
 *   scales[token] = absmax / 127 per row.  Symmetric only (no azp).
 * ref: csrc/quantization/compressed_tensors/int8_quant_kernels.cu:12-22, :50-68, :94-135. */
int mi355x_scaled_mm_int8(void* out, const void* a, const void* b, const float* a_scales,
                          int a_scales_numel, const float* b_scales, int b_scales_numel,
                          const void* bias, float* workspace, int64_t workspace_elems, int m,
                          int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype,
                          mi355x_stream stream);
int mi355x_static_scaled_int8_quant(void* out, const void* input, const float* scale,
                                    int num_tokens, int hidden_size, int64_t input_stride,
                                    int dtype, mi355x_stream stream);
int mi355x_dynamic_scaled_int8_quant(void* out, const void* input, float* scales, int num_tokens,
                                     int hidden_size, int64_t input_stride, int dtype,
                                     mi355x_stream stream);

/* ------------------------------------------------------ decode-step fusions --
 * MI355X-side fusions without a reference op of their own; each is bit-identical to the sequence
 * of reference ops it replaces (tests/test_gpu_w4a16.py, tests/test_gpu_cache_norm_rotary.py).
 *
 * qkv_rope_cache: one launch for what follows the qkv projection of a decode step:
 *   [qkv = T(slab[0] + .. + slab[sk-1]) when sk > 0: the unreduced output of
 *    mi355x_awq_gemm_deferred, slabs [sk][num_tokens][(H + 2 KVH) * D] fp32]
 *   rotary_embedding(positions, q, k), NeoX style, rot_dim == head_size
 *                                        (csrc/pos_encoding_kernels.cu:10-34, :37-100)
 *   reshape_and_cache(k, v, key_cache, value_cache, slot_mapping)   (csrc/cache_kernels.cu:203-255)
 * qkv [num_tokens, qkv_stride] holds q | k | v per row and receives the rotated q and k (and, with
 * slabs, v).  Caches in the x-split layout with x == 8 (2-byte dtypes), heads contiguous inside
 * a block; *_block_stride in elements. */
int mi355x_qkv_rope_cache(void* qkv, int64_t qkv_stride, const float* slabs, int sk,
                          const int64_t* positions, const void* cos_sin_cache, void* key_cache,
                          void* value_cache, const int64_t* slot_mapping, int num_tokens,
                          int num_heads, int num_kv_heads, int head_size, int block_size, int x,
                          int64_t key_block_stride, int64_t value_block_stride, int dtype,
                          mi355x_stream stream);

/* rotary_reshape_and_cache (prefill): rotary_embedding on the KEY rows + reshape_and_cache in one launch — the
 * rotated keys go only to the cache (`key` is not modified; the prefill attention reads K from the cache and can
 * rotate its query rows itself: mi355x_paged_prefill_attention_image).  NeoX style, rot_dim == head_size, x-split
 * cache with x == 8 (2-byte dtypes); otherwise returns 1 (no error): run the two ops.  Same bits in the cache. */
int mi355x_rotary_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                                    const int64_t* slot_mapping, const int64_t* positions,
                                    const void* cos_sin_cache, int num_tokens, int64_t key_stride,
                                    int64_t value_stride, int num_heads, int head_size, int block_size, int x,
                                    int dtype, mi355x_stream stream);

/* rms_norm_image / fused_add_rms_norm_image: rms_norm (fused_add_rms_norm) whose normalised output is written
 * directly as the activation operand image of the prefill GEMM (the format mi355x_awq_gemm_silu_mul_packed
 * produces and mi355x_w4a16_gemm_prepacked / mi355x_awq_gemm_packed_a consume: [row tile of 16][k tile of 32]
 * [64 slots of 8 elements], rows >= num_tokens of the last tile zero), saving the re-tiling launch in front of
 * the qkv / gate_up GEMMs of a prefill chunk.  `input` is NOT modified; fused: residual += input (rounded, as
 * fused_add_rms_norm) in place.  Bit-identical to the row-major ops followed by the re-tiling.  2-byte dtypes,
 * hidden_size 2048 or 4096, num_tokens >= 256; otherwise returns 1 (no error): run the row-major op.
 * image: (roundup(num_tokens, 16) * hidden_size) elements, 16-byte aligned. */
int mi355x_rms_norm_image(void* image, const void* input, const void* weight, float epsilon, int num_tokens,
                          int hidden_size, int64_t input_stride, int dtype, mi355x_stream stream);
int mi355x_fused_add_rms_norm_image(void* image, const void* input, void* residual, const void* weight,
                                    float epsilon, int num_tokens, int hidden_size, int64_t input_stride,
                                    int dtype, mi355x_stream stream);

/* paged_prefill_attention_image: mi355x_paged_prefill_attention whose output [tokens, num_heads * head_size] is
 * written directly as the activation operand image of the prefill GEMM that follows (o_proj): the format of
 * mi355x_rms_norm_image.  The caller zero-fills the last row tile when tokens % 16 != 0.  Same values as the
 * row-major call.  head_size 128, block_size 16, 2-byte dtypes, no sliding window / soft-cap; otherwise returns 1
 * (no error): run mi355x_paged_prefill_attention.
 * positions / cos_sin_cache non-NULL ([tokens] int64, [max_pos, 128] scalar_t): `query` holds the UN-rotated rows
 * and the NeoX rotary (rot_dim 128) is applied while they are loaded — the caller then runs rotary_embedding on
 * the key rows only (they go to the cache); same bits as rotating q first. */
int mi355x_paged_prefill_attention_image(
    void* image, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, int block_size, float scale,
    const int* block_tables, const int* seq_lens, const int* cu_seqlens_q, int max_query_len,
    int max_num_blocks_per_seq, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_cache_dtype, const float* k_scale, const float* v_scale, const int64_t* positions,
    const void* cos_sin_cache, mi355x_stream stream);

/* paged_attention_fused_qkv: qkv_rope_cache (above) folded into the decode attention launch that follows it:
 * the workgroup of (sequence, kv head) builds its query heads, k head and v head of the new token from the
 * qkv row (sk == 0) or its split-K slabs, applies the NeoX rotary to q and k, writes k / v into the cache slot
 * and runs paged_attention_v1 (partition_size 0) or _v2 (partition_size 512 + the reduce launch) on q.
 * Bit-identical to mi355x_qkv_rope_cache + mi355x_paged_attention_v1/_v2 (out and caches; the qkv buffer
 * itself is NOT updated).  positions == cos_sin_cache == NULL: no rotary — q and k are already rotated (the
 * attention-backend form: only mi355x_reshape_and_cache of the new tokens is folded into the decode launch).
 * Applies to 2-byte dtypes, head_size 128, block_size 16, x 8, 4 or 8 query heads per
 * kv head (one workgroup per kv head), scalar_t caches; otherwise returns 1 (no error): run the two calls instead. */
int mi355x_paged_attention_fused_qkv(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const float* slabs, int sk, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int x, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, int64_t kv_block_stride,
    int64_t kv_head_stride, int partition_size, int dtype, mi355x_stream stream);

/* greedy_advance: the bookkeeping between two greedy decode steps, one launch (the reference has no op
 * for it: upstream vLLM's sampler and model runner do it with torch ops — argmax, three in-place adds and
 * the block-table gather that yields slot_mapping; ~13 launches, ~100 us per step at vocab 128256):
 *   tokens[r] = argmax(logits[r, :vocab])              (lowest index among equal maxima)
 *   positions[r] += 1; seq_lens[r] += 1
 *   slot_mapping[r] = block_tables[r][positions[r] / block_size] * block_size + positions[r] % block_size
 * block_tables [num_seqs][max_num_blocks_per_seq] int32, row r = sequence r of the batch. */
int mi355x_greedy_advance(const void* logits, int64_t logits_stride, int num_seqs, int vocab,
                          int64_t* tokens, int64_t* positions, int* seq_lens, int64_t* slot_mapping,
                          const int* block_tables, int max_num_blocks_per_seq, int block_size,
                          int dtype, mi355x_stream stream);

/* ----------------------------------------------------------------- fp8 GEMM --
 * out[M,N] (bf16/f16) = (a_scales . a[M,K] e4m3fn row-major) x
 *                       (b_scales . b[K,N] e4m3fn COLUMN-major, ldb = b.stride(1))
 *                       (+ bias[N]).  a_scales: 1 or M floats; b_scales: 1 or N.
 * `workspace` (4-byte elements, 16-byte aligned, may be NULL): M <= 64: sk*m*n elements let the small-M
 * kernel split K across sk <= 8 workgroups (one partial slab [m, n] each, summed in slab order by a finish
 * kernel: no memset, no atomics — deterministic and HIP-graph-replayable; contents are scratch); M > 320: the prefill
 * kernel (256 x 256 tiles, LDS-DMA ring).  With k % 128 == 0 and 16-byte aligned rows (lda, ldb, a, b) it reads
 * both operands in place and needs no workspace; otherwise (k % 64 == 0) >= (roundup(m,16) + roundup(n,16)) * k
 * bytes let it re-tile the operands into MFMA operand images first; without them the direct kernels run.
 * New capability behind the reference schema cutlass_scaled_mm
 * (csrc/torch_bindings.cpp:251-256; csrc/quantization/cutlass_w8a8/
 *  scaled_mm_entry.cu:34-39,84-140), which the reference only implements for int8. */
int mi355x_scaled_mm_fp8(void* out, const void* a, const void* b, const float* a_scales,
                         int a_scales_numel, const float* b_scales, int b_scales_numel,
                         const void* bias, float* workspace, int64_t workspace_elems, int m,
                         int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype,
                         mi355x_stream stream);

/* Shapes with few 256 x 256 output tiles (chunked-prefill sized m, the narrow n of a TP shard) split K inside the
 * m > 320 kernel when the workspace has room for the partial tiles: mi355x_scaled_mm_split_elems returns the 4-byte
 * elements to ADD to the operand-image scratch above for that (0: this shape is not split).  Without them the GEMM
 * runs unsplit — same bits for int8, fp32 summation order differs for fp8 (deterministic either way). */
int64_t mi355x_scaled_mm_split_elems(int m, int n, int k);

/* Load-time weight image of the 8-bit (fp8 / int8) GEMM's packed path (m > 320): for int8, and for fp8 with
 * k % 128 != 0, mi355x_scaled_mm_* re-tile the weights into 1-KiB operand images on every call (n * k bytes read +
 * written: ~90 us per Llama-3-8B layer); mi355x_scaled_mm_prepack does it once (image: n * k bytes; n % 64 == 0,
 * k % 64 == 0, else returns 1) and mi355x_scaled_mm_prepacked multiplies by the image — bit-identical to the call on
 * b.  workspace: >= roundup(m, 16) * k bytes (the activation image).  fp8 operands with k % 128 == 0 and 16-byte
 * aligned rows need neither: mi355x_scaled_mm_fp8 reads both in place at the image's speed (such weights gain
 * nothing from an image, and mi355x_scaled_mm_prepacked then takes workspace == NULL).  Decode-sized calls keep
 * streaming b itself.  (No reference op: the reference's int8 path repacks nothing, cutlass reads b directly.) */
int mi355x_scaled_mm_prepack(void* image, const void* b, int n, int k, int64_t ldb, mi355x_stream stream);
int mi355x_scaled_mm_prepacked(void* out, const void* a, const void* b_image, const float* a_scales,
                               int a_scales_numel, const float* b_scales, int b_scales_numel, const void* bias,
                               float* workspace, int64_t workspace_elems, int m, int n, int k, int64_t lda,
                               int64_t ldc, int out_dtype, int is_int8, mi355x_stream stream);

/* ---- fp8 decode step: the K split of a scaled GEMM reduced by its consumer (round 3; no reference ops) --------
 * Decode-sized (m <= 64) W8A8 GEMMs are launch-latency-bound: every launch of the fp8 decoder layer
 * costs ~4.7 us whatever it does.  mi355x_scaled_mm_fp8_deferred is mi355x_scaled_mm_fp8 (no bias) that leaves a K
 * split as fp32 partial slabs workspace[sk][m][n] (*sk_out = sk > 1) instead of launching its finish kernel;
 * *sk_out = 0: `out` is final.  The consumers below then compute T(sum of slabs * a_scale[row] * b_scale[col]) —
 * the finish kernel's bits — on their way in; each is tested bit for bit against the unfused op sequence. */
int mi355x_scaled_mm_fp8_deferred(void* out, const void* a, const void* b, const float* a_scales,
                                  int a_scales_numel, const float* b_scales, int b_scales_numel,
                                  float* workspace, int64_t workspace_elems, int m, int n, int k,
                                  int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int* sk_out,
                                  mi355x_stream stream);
/* mi355x_paged_attention_fused_qkv whose qkv slabs come from an fp8 GEMM (a_scales [num_seqs] or [1], b_scales
 * [(num_heads + 2 num_kv_heads) * head_size] or [1]; b_scales NULL = plain slabs / row), and — out_q != NULL,
 * partitioned form only, <= 16 heads, <= 64 partitions — whose reduce launch also quantises the attention output
 * per token (out_q e4m3 [num_seqs, num_heads * head_size], out_scales float [num_seqs]: the bits of
 * dynamic_per_token_scaled_fp8_quant on the reduce's output, the input of an fp8 o_proj; `out` is then unused).
 * Returns 1 when not applicable. */
int mi355x_paged_attention_fused_qkv_w8(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv, int64_t qkv_stride,
    const float* slabs, int sk, const int64_t* positions, const void* cos_sin_cache,
    const int64_t* slot_mapping, void* key_cache, void* value_cache, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int x, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int max_seq_len, int64_t kv_block_stride,
    int64_t kv_head_stride, int partition_size, int dtype, const float* a_scales, int a_scales_numel,
    const float* b_scales, int b_scales_numel, void* out_q, float* out_scales, mi355x_stream stream);
/* silu_and_mul_per_token_quant on the slabs [sk][num_tokens][2 d] of an fp8 gate_up GEMM.  Returns 1 when not
 * applicable (d % 8, d > 16384, alignment). */
int mi355x_silu_and_mul_per_token_quant_slabs(void* out, float* scales, const float* slabs, int sk,
                                              const float* a_scales, int a_scales_numel,
                                              const float* b_scales, int b_scales_numel, int num_tokens, int d,
                                              int dtype, mi355x_stream stream);
/* rms_norm_dynamic_per_token_quant whose input rows are the slabs [sk][num_tokens][hidden_size] of an fp8 GEMM
 * (o_proj / down_proj at TP = 1); residual (fused add) as in the op itself. */
int mi355x_rms_norm_dynamic_per_token_quant_slabs(void* out, const float* slabs, int sk,
                                                  const float* a_scales, int a_scales_numel,
                                                  const float* b_scales, int b_scales_numel,
                                                  const void* weight, float* scales, float epsilon,
                                                  const float* scale_ub, void* residual, int num_tokens,
                                                  int hidden_size, int dtype, mi355x_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_HOTPATH_H_ */
