/*
 * swiftllm_hip.h — C ABI of libswiftllm_hip.so
 *
 * The drop-in boundary for swiftLLM's data plane (LlamaModel.forward) on MI355X / gfx950.
 * Every entry point replaces one operator of the reference's Python/Triton operator layer
 * (the modules under swiftllm/worker/kernels) or its one native function (csrc/src/block_swapping.cpp).
 * The reference file:line each entry replaces is cited above its declaration.
 *
 * Conventions
 *   - plain `extern "C"`, device pointers as void*, sizes as int32/int64, no torch types;
 *   - `dtype`: SWL_F16 (the reference's hard-coded precision) or SWL_BF16 (headline precision);
 *   - `stream` is a hipStream_t passed as void*; NOTHING is ever launched on the null stream
 *     unless the caller passes NULL explicitly. No allocation, no synchronisation, no global
 *     state: every call is re-entrant and may be captured into a hipGraph;
 *   - returns SWL_OK (0) or a negative SWL_ERR_* code; the Python shim maps non-zero to
 *     RuntimeError (the reference surfaces failures as AssertionError/RuntimeError);
 *   - zero-size batches are legal everywhere and return SWL_OK without launching
 *     (the reference's idle engine spins on empty forwards: server/engine.py:115-171);
 *   - all offsets into the KV pools are 64-bit (288 GB pools exceed 2^31 elements).
 *
 * KV pool layout (reference model.py:138-148): [num_blocks, num_layers, num_kv_heads, block_size, head_dim],
 * contiguous, one pool for K and one for V.
 */
#ifndef SWIFTLLM_HIP_H
#define SWIFTLLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWL_ABI_VERSION 1

#define SWL_F16 0
#define SWL_BF16 1

#define SWL_OK 0
#define SWL_ERR_BAD_ARG (-1)     /* null pointer, negative size, misaligned pointer/stride          */
#define SWL_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are specialised for              */
#define SWL_ERR_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch                */
#define SWL_ERR_RUNTIME (-4)     /* a HIP runtime call (memcpy) failed                              */

typedef void *swl_stream_t; /* hipStream_t */

int swl_abi_version(void);
/* Human-readable name for an SWL_ERR_* code. */
const char *swl_strerror(int code);

/* ---- RMSNorm ------------------------------------------------------------------------------
 * reference: rmsnorm.py:5-37 (_fwd_rmsnorm / rmsnorm_inplace)
 * x[T, hidden] <- x * rsqrt(mean(x^2) + eps) * w      (fp32 math, one rounding to dtype)
 * hidden % 8 == 0, hidden <= 16384; rows contiguous (row stride == hidden). */
int swl_rmsnorm(void *x, const void *w, float eps, int64_t num_tokens, int32_t hidden,
                int32_t dtype, swl_stream_t stream);

/* reference: rmsnorm.py:39-89 (_fwd_fused_add_rmsnorm / fused_add_rmsnorm_inplace)
 * residual <- x + residual (rounded to dtype and stored); x <- rmsnorm(residual) * w. */
int swl_fused_add_rmsnorm(void *x, void *residual, const void *w, float eps, int64_t num_tokens,
                          int32_t hidden, int32_t dtype, swl_stream_t stream);

/* ---- Rotary embedding ---------------------------------------------------------------------
 * reference: rotary_emb.py:7-58 (_fwd_rotary_embedding / rotary_embedding_inplace)
 * rotate-half RoPE in place on q[T, H, D] and k[T, KVH, D]; every multiply/add is rounded to
 * dtype as in the reference. cos/sin are [rows, D/2] in dtype. If pos_idx != NULL the row used
 * for token t is pos_idx[t] (tables = the model's full rope cache, model.py:224-225); if NULL
 * the row is t (tables already gathered: model.py:350-351).
 * q_tok_stride / k_tok_stride: elements between consecutive tokens (>= H*D / KVH*D; lets q,k be
 * column slices of one fused qkv GEMM output). D in {32, 64, 128, 256}. */
int swl_rotary(void *q, void *k, const void *cos_table, const void *sin_table,
               const int32_t *pos_idx, int64_t num_tokens, int32_t num_q_heads,
               int32_t num_kv_heads, int32_t head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
               int32_t dtype, swl_stream_t stream);

/* ---- KV-cache store -----------------------------------------------------------------------
 * reference: kvcache_mgmt.py:10-48 (_fwd_kvcache_mgmt_prefill_kernel), launcher :81-109
 * For prefill seq s (id seq_ids[s], tokens [start_locs[s], +seq_lens[s]) of k/v), logical block j:
 * pool[block_table[seq_id, j], layer, kvh, t % bs, :] = k[start + j*bs + t, kvh, :]   (bit-exact). */
int swl_store_kv_prefill(void *k_cache, void *v_cache, const void *k, const void *v,
                         const int32_t *block_table, const int32_t *seq_ids,
                         const int32_t *start_locs, const int32_t *seq_lens,
                         int32_t num_prefill_seqs, int32_t max_prefill_len, int32_t cur_layer,
                         int32_t num_layers, int32_t num_kv_heads, int32_t block_size,
                         int32_t head_dim, int32_t max_blocks_per_seq, int64_t k_tok_stride,
                         int64_t v_tok_stride, int32_t dtype, swl_stream_t stream);

/* Prefill fusion: swl_rotary on the prefill tokens' q and k (rotary_emb.py:7-42) + swl_store_kv_prefill
 * (kvcache_mgmt.py:10-48) in ONE pass over k — transformer_layer.py:62-77 for the prefill sequences of a forward. q and k
 * are rotated in place (k is also what the prefill attention reads), the rotated k and v go into the pools. pos_idx: rope
 * table row per token (NULL: the token's index inside its sequence). Bit-identical to the two calls it replaces. */
int swl_rotary_store_kv_prefill(void *q, void *k, const void *v, const void *cos_table, const void *sin_table,
                                const int32_t *pos_idx, void *k_cache, void *v_cache, const int32_t *block_table,
                                const int32_t *seq_ids, const int32_t *start_locs, const int32_t *seq_lens,
                                int32_t num_prefill_seqs, int32_t max_prefill_len, int32_t cur_layer, int32_t num_layers,
                                int32_t num_q_heads, int32_t num_kv_heads, int32_t block_size, int32_t head_dim,
                                int32_t max_blocks_per_seq, int64_t q_tok_stride, int64_t k_tok_stride,
                                int64_t v_tok_stride, int32_t dtype, swl_stream_t stream);
/* reference: kvcache_mgmt.py:50-79 (_fwd_kvcache_mgmt_decoding_kernel), launcher :111-122
 * For decoding seq i (length len_i INCLUDING the new token): token i of k/v goes to slot
 * (block_table[seq_id, (len-1)/bs], (len-1)%bs). */
int swl_store_kv_decode(void *k_cache, void *v_cache, const void *k, const void *v,
                        const int32_t *block_table, const int32_t *seq_ids,
                        const int32_t *seq_lens, int32_t num_decoding_seqs, int32_t cur_layer,
                        int32_t num_layers, int32_t num_kv_heads, int32_t block_size,
                        int32_t head_dim, int32_t max_blocks_per_seq, int64_t k_tok_stride,
                        int64_t v_tok_stride, int32_t dtype, swl_stream_t stream);

/* ---- SiLU-gate ----------------------------------------------------------------------------
 * reference: silu_and_mul.py:5-34 (_fwd_silu_and_mul / silu_and_mul_inplace)
 * x[T, 2*I]: x[:, :I] <- x[:, :I] * round(silu_fp32(x[:, I:]))   (up = first half, gate = second,
 * weight.py:133).  I % 8 == 0 (the reference needs I % 256 == 0). */
int swl_silu_mul(void *x, int64_t num_tokens, int32_t ffn_inter_dim, int32_t dtype,
                 swl_stream_t stream);

/* Decode attention fed directly by the split-K slabs of the fused qkv projection (swl_gemm_skinny_partial,
 * k_splits >= 1): the rotary embedding and the KV-cache store of the new token (reference rotary_emb.py:7-42 +
 * kvcache_mgmt.py:50-79, separate launches at transformer_layer.py:62-77) run in the attention kernel's
 * prologue; results are bit-identical to swl_splitk_rotary_store_kv_decode + swl_paged_attn_decode.
 * qkv_slabs: [k_splits][Bd][(H + 2*KVH) * D] fp32. Other arguments as swl_paged_attn_decode. */
int swl_paged_attn_decode_qkv(void *o, const float *qkv_slabs, int32_t k_splits, const void *cos_table,
                              const void *sin_table, const int32_t *pos_idx, void *k_cache, void *v_cache,
                              const int32_t *block_table, const int32_t *seq_ids, const int32_t *seq_lens,
                              void *scratch, float softmax_scale, int32_t num_decoding_seqs,
                              int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t num_layers,
                              int32_t block_size, int32_t cur_layer, int32_t max_blocks_per_seq,
                              int32_t seq_block_size, int32_t num_seq_blocks, int64_t o_tok_stride,
                              int32_t dtype, swl_stream_t stream);

/* ---- Greedy sampling --------------------------------------------------------------------------
 * reference: post_layer.py:40 (`torch.argmax(logits, dim=1)`)
 * out[r] = argmax_j x[r, j] (int64), ties -> lowest j, NaNs never selected. n % 8 == 0.
 * scratch: swl_argmax_scratch_bytes(num_rows) bytes, 16-byte aligned. */
size_t swl_argmax_scratch_bytes(int64_t num_rows);
int swl_argmax(int64_t *out, const void *x, void *scratch, size_t scratch_bytes, int64_t num_rows, int32_t n,
               int64_t row_stride, int32_t dtype, swl_stream_t stream);

/* ---- Paged attention, decode (flash-decoding, "paged attention v2") --------------------------
 * reference: paged_attn.py:9-108 (phase 1), :111-149 (phase 2), launcher :152-222
 * q[Bd, H, D] (token stride q_tok_stride), o[Bd, H, D] (token stride o_tok_stride).
 * Phase 1 writes per (seq, q-head, seq-block) a normalised partial mid_o (fp32) and the base-2
 * log-sum-exp mid_lse (fp32), phase 2 merges them; with num_seq_blocks == 1 phase 1 writes o
 * directly. Scratch: mid_o [Bd, H, num_seq_blocks, D] fp32 followed by mid_lse [Bd, H, num_seq_blocks]
 * fp32 — the same shapes the reference allocates at paged_attn.py:170-180.
 * One workgroup serves ALL G = H/KVH q-heads of a kv-head, so every KV byte is fetched once.
 * seq_block_size % block_size == 0; block_size == 16; D in {32, 64, 128}; G in {1, 2, 4, 8}. */
size_t swl_paged_attn_scratch_bytes(int32_t num_decoding_seqs, int32_t num_q_heads,
                                    int32_t head_dim, int32_t num_seq_blocks);

int swl_paged_attn_decode(void *o, const void *q, const void *k_cache, const void *v_cache,
                          const int32_t *block_table, const int32_t *seq_ids,
                          const int32_t *seq_lens, void *scratch, float softmax_scale,
                          int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads,
                          int32_t head_dim, int32_t num_layers, int32_t block_size,
                          int32_t cur_layer, int32_t max_blocks_per_seq, int32_t seq_block_size,
                          int32_t num_seq_blocks, int64_t q_tok_stride, int64_t o_tok_stride,
                          int32_t dtype, swl_stream_t stream);

/* Phase 1 / phase 2 separately (parity tests compare mid_o / mid_lse with the reference's). */
int swl_paged_attn_phase1(void *o_direct, const void *q, const void *k_cache, const void *v_cache,
                          const int32_t *block_table, const int32_t *seq_ids,
                          const int32_t *seq_lens, float *mid_o, float *mid_lse,
                          float softmax_scale, int32_t num_decoding_seqs, int32_t num_q_heads,
                          int32_t num_kv_heads, int32_t head_dim, int32_t num_layers,
                          int32_t block_size, int32_t cur_layer, int32_t max_blocks_per_seq,
                          int32_t seq_block_size, int32_t num_seq_blocks, int64_t q_tok_stride,
                          int64_t o_tok_stride, int32_t dtype, swl_stream_t stream);
int swl_paged_attn_phase2(void *o, const float *mid_o, const float *mid_lse,
                          const int32_t *seq_lens, int32_t num_decoding_seqs, int32_t num_q_heads,
                          int32_t head_dim, int32_t seq_block_size, int32_t num_seq_blocks,
                          int64_t o_tok_stride, int32_t dtype, swl_stream_t stream);

/* ---- Prefill attention (varlen causal flash attention, GQA) -------------------------------------
 * reference: the call at transformer_layer.py:83-96 (vllm_flash_attn.flash_attn_varlen_func) and its
 * in-repo equivalent prefill_attn.py:9-139 (_fwd_prefill_attention / prefill_attention).
 * q[P, H, D], k/v[P, KVH, D] are the FRESH projections (the paged pool is not read), cu_seqlens is
 * prefill_seq_start_locs_with_end (int32 [Bp+1]); o[P, H, D]. MFMA 32x32x16, fp32 softmax.
 * D in {32, 64, 128}; all four tensors 16-byte aligned with token strides that are multiples of 8 elements. */
int swl_prefill_attn_varlen(void *o, const void *q, const void *k, const void *v,
                            const int32_t *cu_seqlens, int32_t num_prefill_seqs,
                            int32_t max_prefill_len, int32_t num_q_heads, int32_t num_kv_heads,
                            int32_t head_dim, float softmax_scale, int64_t q_tok_stride,
                            int64_t k_tok_stride, int64_t v_tok_stride, int64_t o_tok_stride,
                            int32_t dtype, swl_stream_t stream);

/* ---- Block-table maintenance ------------------------------------------------------------------
 * reference: block_mgmt.py:5-46 (set), :49-80 (unset), :83-127 (gather + unset)
 * block_table int32 [max_seqs, max_blocks_per_seq]; num_seq_allocated_blocks int32 [max_seqs];
 * is_block_free uint8(bool) [num_blocks].
 * set: for batch entry i (seq s): block_table[s, n_s : n_s + need_i] = candidates[off_i : off_i + need_i],
 *      n_s += need_i, where off = exclusive cumsum(block_needed) (passed in by the caller).
 *      If is_block_free != NULL the candidates are also marked used (is_block_free[c] = 0), which
 *      fuses the reference's separate scatter at block_manager.py:52. */
int swl_block_table_set(int32_t *num_seq_allocated_blocks, int32_t *block_table,
                        const int32_t *candidate_blocks, const int32_t *seq_ids,
                        const int32_t *block_needed, const int32_t *block_needed_excl_cumsum,
                        uint8_t *is_block_free, int32_t batch_size, int32_t max_blocks_per_seq,
                        swl_stream_t stream);
/* unset: is_block_free[block_table[s, 0:n_s]] = 1; n_s = 0. */
int swl_block_table_unset(int32_t *num_seq_allocated_blocks, const int32_t *block_table,
                          const int32_t *seq_ids, uint8_t *is_block_free, int32_t batch_size,
                          int32_t max_blocks_per_seq, swl_stream_t stream);
/* gather: out[off_i : off_i + n_s] = block_table[s, 0:n_s]; then unset. */
int swl_block_table_gather(int32_t *num_seq_allocated_blocks, const int32_t *block_table,
                           const int32_t *seq_ids, uint8_t *is_block_free,
                           const int32_t *out_excl_cumsum, int32_t *gathered_block_ids,
                           int32_t batch_size, int32_t max_blocks_per_seq, swl_stream_t stream);

/* ---- Block swapping (host code) ---------------------------------------------------------------
 * reference: csrc/src/block_swapping.cpp:22-85 (swap_blocks), binding csrc/src/entrypoints.cpp:5-7
 * Copies block src_ids[i] -> dst_ids[i] for K and V between the GPU pools and the host swap pools,
 * run-length-coalescing consecutive (src, dst) pairs into one hipMemcpyAsync each, on `stream`.
 * src_ids/dst_ids are HOST arrays. block_bytes = bytes of one block in one pool. */
int swl_swap_blocks(const int64_t *src_ids, const int64_t *dst_ids, int64_t num_blocks_to_swap,
                    int32_t is_swap_in, void *k_cache, void *v_cache, void *k_swap, void *v_swap,
                    int64_t block_bytes, swl_stream_t stream);

/* ---- Fused hot-path helpers (no reference twin; result-identical compositions) ------------------
 * Decode-only: rotary on q,k followed by the decode KV store of the rotated k and of v, in one launch
 * (= swl_rotary(pos_idx) + swl_store_kv_decode). */
int swl_rotary_store_kv_decode(void *q, void *k, const void *v, const void *cos_table,
                               const void *sin_table, const int32_t *pos_idx, void *k_cache,
                               void *v_cache, const int32_t *block_table, const int32_t *seq_ids,
                               const int32_t *seq_lens, int32_t num_decoding_seqs,
                               int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                               int32_t cur_layer, int32_t num_layers, int32_t block_size,
                               int32_t max_blocks_per_seq, int64_t q_tok_stride,
                               int64_t k_tok_stride, int64_t v_tok_stride, int32_t dtype,
                               swl_stream_t stream);

/* ---- Skinny GEMM for decode batches -----------------------------------------------------------------
 * reference: kernels/linear.py:3-12 (F.linear), call sites transformer_layer.py:54-56,117,126,128 and
 * post_layer.py:38. out[M, N] = x[M, K] . W[N, K]^T with fp32 accumulation and one rounding, for
 * M <= 32 tokens: a weight-streaming MFMA kernel (every byte of W read once, in full lines). When N/32
 * tiles cannot fill the chip, K is split across workgroups into fp32 partial slabs in `workspace` and
 * reduced in a fixed order by a second kernel. N % 32 == 0, K % 128 == 0; x/out rows may be strided
 * (elements). k_splits: 0 = choose, or a power of two <= 16. Larger M: use the BLAS.
 * workspace: >= swl_gemm_skinny_workspace_bytes(M, N, K) bytes (0 = none needed). */
size_t swl_gemm_skinny_workspace_bytes(int32_t M, int32_t N, int32_t K);
int swl_gemm_skinny(void *out, const void *x, const void *w, void *workspace, size_t workspace_bytes,
                    int32_t M, int32_t N, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                    int32_t k_splits, int32_t dtype, swl_stream_t stream);

/* FFN up/gate projection with the SiLU-gate fused into the epilogue (reference: transformer_layer.py:126-127
 * = linear + silu_and_mul_inplace): out[M, I] = (x . W[0:I]^T) * silu(x . W[I:2I]^T), W = [up ; gate]
 * (weight.py:133). Bit-identical to swl_gemm_skinny(k_splits = 1) + swl_silu_mul. I % 32 == 0,
 * K % 128 == 0, M <= 32. */
int swl_gemm_skinny_silu_gate(void *out, const void *x, const void *w_up_gate, int32_t M, int32_t I,
                              int32_t K, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                              swl_stream_t stream);

/* ---- Pre-packed weights -----------------------------------------------------------------------------------------
 * swl_gemm_pack_weight repacks W[N, K] once (load time) into MFMA-fragment order [N/32][K/16][64 lanes][8 elements]:
 * the A operand of a 32x32x16 MFMA becomes one contiguous KiB and a wave's K range for its 32 rows one sequential
 * run, so the decode GEMMs read W with long DRAM bursts and without an LDS transpose (6.0-6.3 instead of 5.3-5.8
 * TB/s). The *_packed entry points take the packed copy and otherwise behave exactly like their row-major twins
 * (same arguments, same bits). N % 32 == 0, K % 128 == 0 (pack: K % 16 == 0), M <= 32. */
int swl_gemm_pack_weight(void *dst, const void *src, int32_t N, int32_t K, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed(void *out, const void *x, const void *w_packed, void *workspace, size_t workspace_bytes,
                           int32_t M, int32_t N, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                           int32_t k_splits, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed, int32_t M,
                                   int32_t N, int32_t K, int64_t x_row_stride, int32_t k_splits, int32_t dtype,
                                   swl_stream_t stream);
int swl_gemm_skinny_packed_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M, int32_t I,
                                     int32_t K, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                                     swl_stream_t stream);
/* Medium batches on a packed weight: out[M, N] = x . W^T for 32 < M <= 64 tokens (valid for any M <= 64); 2 blocks of
 * 32 tokens share every weight fragment (65..256 tokens: swl_gemm_packed_wide below). workspace >= k_splits * M * N * 4 bytes when K is split
 * (k_splits = 0: library's choice; 16 * M * N * 4 bytes cover any). */
int swl_gemm_packed_mid(void *out, const void *x, const void *w_packed, void *workspace, size_t workspace_bytes,
                        int32_t M, int32_t N, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                        int32_t k_splits, int32_t dtype, swl_stream_t stream);
int swl_gemm_packed_mid_choose_splits(int32_t M, int32_t N, int32_t K); /* its choice for k_splits = 0 */
/* ... with the SiLU-gate of the FFN in its epilogue: out[M, I] = up * silu(gate), W = [up ; gate] packed */
int swl_gemm_packed_mid_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M, int32_t I,
                                  int32_t K, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                                  swl_stream_t stream);
/* Large decode batches on a packed weight (csrc/gemm_wide.hip): out[M, N] = x . W^T for up to 256 tokens — the projections
 * of reference kernels/linear.py:3-12 (transformer_layer.py:54-56,117,126,128) at the batch sizes a 264 GB KV pool holds;
 * every weight fragment feeds ceil(M/32) <= 8 MFMAs, x^T shared by the workgroup through LDS. N % 32 == 0, K % 64 == 0.
 * waves_per_group: 0 = library's choice, 4 or 8; k_splits: 0 = library's choice, else a power of two <= 16 with
 * K % (64 * k_splits) == 0; workspace >= k_splits * M * N * 4 bytes when K is split. */
int swl_gemm_packed_wide(void *out, const void *x, const void *w_packed, void *workspace, size_t workspace_bytes,
                         int32_t M, int32_t N, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                         int32_t waves_per_group, int32_t k_splits, int32_t dtype, swl_stream_t stream);
size_t swl_gemm_packed_wide_workspace_bytes(int32_t M, int32_t N, int32_t K); /* for the library's own plan */
int swl_gemm_packed_wide_choose_splits(int32_t M, int32_t N, int32_t K);       /* its choice for k_splits = 0 */
/* ... stopping at the fp32 partial slabs [k_splits][M][N] (k_splits >= 1) for the split-K consumers */
int swl_gemm_packed_wide_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed, int32_t M,
                                 int32_t N, int32_t K, int64_t x_row_stride, int32_t waves_per_group, int32_t k_splits,
                                 int32_t dtype, swl_stream_t stream);
/* ... with the SiLU-gate of the FFN (reference kernels/silu_and_mul.py:5-34) in its epilogue: out[M, I] = up * silu(gate) */
int swl_gemm_packed_wide_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M, int32_t I,
                                   int32_t K, int64_t x_row_stride, int64_t out_row_stride, int32_t waves_per_group,
                                   int32_t dtype, swl_stream_t stream);
/* ... stopping at the fp32 partial slabs [k_splits][M][N] (k_splits >= 1) for the split-K consumers */
int swl_gemm_packed_mid_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed, int32_t M,
                                int32_t N, int32_t K, int64_t x_row_stride, int32_t k_splits, int32_t dtype,
                                swl_stream_t stream);

/* Split-K without the reduce launch: the GEMM stops at its fp32 partial slabs [k_splits][M][N] and a FUSED
 * CONSUMER adds them (slab order, one rounding — bit-identical to swl_gemm_skinny's own reduce):
 *   o_proj / down_proj  -> swl_splitk_fused_add_rmsnorm      (reference: rmsnorm.py:39-89)
 *   fused qkv           -> swl_splitk_rotary_store_kv_decode (reference: rotary_emb.py + kvcache_mgmt.py:50-79)
 *   anything else       -> swl_splitk_reduce */
int swl_gemm_skinny_choose_splits(int32_t N, int32_t K);
/* ... on a packed weight (swl_gemm_skinny_packed*, k_splits = 0): splits may differ by one 128-column tile when K has no
 * power-of-two split of whole tiles that fills the chip (Llama-2-7B down_proj: K = 11008 -> 8 splits of 10-11 tiles). */
int swl_gemm_skinny_packed_choose_splits(int32_t N, int32_t K); /* 0 = shape unsupported, 1 = no split */
int swl_gemm_skinny_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w, int32_t M,
                            int32_t N, int32_t K, int64_t x_row_stride, int32_t k_splits,
                            int32_t dtype, swl_stream_t stream);
int swl_splitk_reduce(void *out, const float *slabs, int32_t k_splits, int32_t M, int32_t N,
                      int64_t out_row_stride, int32_t dtype, swl_stream_t stream);
/* residual <- round(sum slabs) + residual ; x_out <- rmsnorm(residual) * w   (slabs: [k_splits][T][hidden]) */
int swl_splitk_fused_add_rmsnorm(void *x_out, void *residual, const void *w, float eps,
                                 const float *slabs, int32_t k_splits, int64_t num_tokens,
                                 int32_t hidden, int32_t dtype, swl_stream_t stream);
/* qkv_slabs: [k_splits][Bd][(H+2*KVH)*D]. Writes rotated q, rotated k and v to the (output) buffers
 * q_out/k_out/v_out and rotated k, v into the paged pools — swl_rotary_store_kv_decode on reduced inputs. */
int swl_splitk_rotary_store_kv_decode(void *q_out, void *k_out, void *v_out, const float *qkv_slabs,
                                      int32_t k_splits, const void *cos_table, const void *sin_table,
                                      const int32_t *pos_idx, void *k_cache, void *v_cache,
                                      const int32_t *block_table, const int32_t *seq_ids,
                                      const int32_t *seq_lens, int32_t num_decoding_seqs,
                                      int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                      int32_t cur_layer, int32_t num_layers, int32_t block_size,
                                      int32_t max_blocks_per_seq, int64_t q_tok_stride,
                                      int64_t k_tok_stride, int64_t v_tok_stride, int32_t dtype,
                                      swl_stream_t stream);

/* Per-step decode metadata derived on the device (so a captured hipGraph can be replayed):
 * pos_idx[i] = seq_lens[i] - 1. */
int swl_decode_positions(int32_t *pos_idx, const int32_t *seq_lens, int32_t num_decoding_seqs,
                         swl_stream_t stream);

/* ---- deferred RMSNorm on the decode fast path ---------------------------------------------------------------------
 * fused_add_rmsnorm (reference rmsnorm.py:67-89, called at transformer_layer.py:46,120) needs a whole token row before it
 * can write anything, so as a split-K consumer it runs one workgroup per token (32 workgroups on 256 CUs). The 1/rms is
 * a per-row scalar that commutes with the projection that follows, so the element-wise part runs fully parallel here and
 * the consumers below apply the scale in fp32 before their one rounding:
 *   swl_splitk_add_scale: residual[t, :] += round(sum_k slabs[k][t, :]) (stored, rounded to the storage dtype, as the
 *     reference stores it); x_scaled = round(residual * w); ssq_out[hidden / 1024][num_tokens] = per-1024-column sums of
 *     squares of the updated residual rows. hidden % 1024 == 0.
 *   swl_gemm_skinny_packed_silu_gate_rs: swl_gemm_skinny_packed_silu_gate of rstd[m] * (x_scaled . [up ; gate]^T),
 *     rstd[m] = 1/sqrt(sum_p row_ssq[p][m] / K + eps). Replaces transformer_layer.py:120-127 for decode batches.
 *   swl_paged_attn_decode_qkv_rs: swl_paged_attn_decode_qkv whose fused-qkv slab sums are scaled the same way before they
 *     are rounded, rotated and stored (transformer_layer.py:46-77 + paged_attn.py:152-222). k_splits in {1, 2, 4}.
 * ssq_parts <= 8 everywhere. */
int swl_splitk_add_scale(void *x_scaled, void *residual, const void *w, const float *slabs, int32_t k_splits,
                         float *ssq_out, int64_t num_tokens, int32_t hidden, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_silu_gate_rs(void *out, const void *x, const void *w_up_gate_packed, const float *row_ssq,
                                        int32_t ssq_parts, float eps, int32_t M, int32_t I, int32_t K,
                                        int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, swl_stream_t stream);
int swl_paged_attn_decode_qkv_rs(void *o, const float *qkv_slabs, int32_t k_splits, const float *row_ssq,
                                 int32_t ssq_parts, int32_t hidden, float eps, const void *cos_table,
                                 const void *sin_table, const int32_t *pos_idx, void *k_cache, void *v_cache,
                                 const int32_t *block_table, const int32_t *seq_ids, const int32_t *seq_lens,
                                 void *scratch, float softmax_scale, int32_t num_decoding_seqs, int32_t num_q_heads,
                                 int32_t num_kv_heads, int32_t head_dim, int32_t num_layers, int32_t block_size,
                                 int32_t cur_layer, int32_t max_blocks_per_seq, int32_t seq_block_size,
                                 int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype, swl_stream_t stream);

/* ---- decode projections for very small batches that consume the previous projection's slabs (csrc/gemm_tiny.hip) ----
 * M <= swl_gemm_tiny_max_tokens() (4). At batch 1 the two split-K consumers of a decode layer (swl_splitk_add_scale:
 * reference rmsnorm.py:67-89 at transformer_layer.py:46,120) are 10 % of the layer and run on four workgroups; here the
 * NEXT projection rebuilds its activations itself while its first weight tiles are in flight — residual_out = round(sum
 * slabs_in) + residual_in (stored by the workgroups with blockIdx.x == 0; residual_out != residual_in, the others still
 * read the old rows), x = round(residual_out * norm_w) kept in LDS for the whole K loop — and carries the sums of squares
 * to where the deferred 1/rms is applied. Packed weights (swl_gemm_pack_weight), K % 128 == 0, K / k_splits_out <= 4096.
 *   swl_gemm_tiny_partial_from_splitk: the fused qkv projection (transformer_layer.py:46-56): slabs_out[k_splits_out][M][N]
 *     fp32 = x . W^T per K-chunk (the bits of swl_splitk_add_scale + swl_gemm_skinny_packed_partial), ssq_out[k_splits_out]
 *     [M] = sums of squares of the residual rows per K-chunk — what swl_paged_attn_decode_qkv_rs takes as row_ssq
 *     (ssq_parts = k_splits_out). slabs_out != slabs_in.
 *   swl_gemm_tiny_silu_gate_from_splitk: the FFN up/gate projection + SiLU-gate (transformer_layer.py:120-127):
 *     out[M, I] = up * silu(gate) of rstd[m] * (x . [up ; gate]^T), rstd from the same rows. I % 64 == 0, K <= 4096. */
/*   swl_paged_attn_decode_qkv_rs_partials: swl_paged_attn_decode_qkv_rs stopped after phase 1 when num_seq_blocks > 1 — the
 *     flash-decoding partials stay in `scratch` (mid_o fp32 [Bd][H][nsb][D], then mid_lse fp32 [Bd][H][nsb]; reference
 *     paged_attn.py:106-108), `o` may be NULL; with num_seq_blocks == 1 it is swl_paged_attn_decode_qkv_rs.
 *   swl_gemm_tiny_partial_from_attn: o_proj (transformer_layer.py:117) on those partials — every workgroup merges the
 *     partials of its K-chunk of heads itself (phase 2's LSE-weighted sum, paged_attn.py:108-150, rounded to the storage
 *     dtype as phase 2 stores it) and writes slabs_out[k_splits_out][M][N]; K = num_q_heads * head_dim. Replaces the
 *     phase-2 launch of a batch of <= 4 sequences. */
int swl_paged_attn_decode_qkv_rs_partials(void *o, const float *qkv_slabs, int32_t k_splits, const float *row_ssq,
                                          int32_t ssq_parts, int32_t hidden, float eps, const void *cos_table,
                                          const void *sin_table, const int32_t *pos_idx, void *k_cache, void *v_cache,
                                          const int32_t *block_table, const int32_t *seq_ids, const int32_t *seq_lens,
                                          void *scratch, float softmax_scale, int32_t num_decoding_seqs,
                                          int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                          int32_t num_layers, int32_t block_size, int32_t cur_layer,
                                          int32_t max_blocks_per_seq, int32_t seq_block_size, int32_t num_seq_blocks,
                                          int64_t o_tok_stride, int32_t dtype, swl_stream_t stream);
int swl_gemm_tiny_partial_from_attn(float *slabs_out, size_t slabs_out_bytes, int32_t k_splits_out,
                                    const float *attn_scratch, const int32_t *seq_lens, int32_t num_q_heads,
                                    int32_t head_dim, int32_t seq_block_size, int32_t num_seq_blocks,
                                    const void *w_packed, int32_t M, int32_t N, int32_t dtype, swl_stream_t stream);
int swl_gemm_tiny_max_tokens(void);
int swl_gemm_tiny_partial_from_splitk(float *slabs_out, size_t slabs_out_bytes, int32_t k_splits_out, float *ssq_out,
                                      const float *slabs_in, int32_t k_splits_in, const void *residual_in,
                                      void *residual_out, const void *norm_w, const void *w_packed, int32_t M, int32_t N,
                                      int32_t K, int32_t dtype, swl_stream_t stream);
int swl_gemm_tiny_silu_gate_from_splitk(void *out, const float *slabs_in, int32_t k_splits_in, const void *residual_in,
                                        void *residual_out, const void *norm_w, float eps, const void *w_up_gate_packed,
                                        int32_t M, int32_t I, int32_t K, int64_t out_row_stride, int32_t dtype,
                                        swl_stream_t stream);

/* ---- hidden-wide projections of a decode layer with K split INSIDE the workgroup (csrc/gemm_rows.hip) ----------------------
 * o_proj / down_proj (reference kernels/linear.py:3-12 at layers/transformer_layer.py:117,128) and the residual add that
 * follows them (rmsnorm.py:54-57 at transformer_layer.py:120 and the next layer's :46) in ONE launch, M <= 32 tokens, packed
 * weight (swl_gemm_pack_weight), K % 1024 == 0, N % 32 == 0. A workgroup owns 16 rows of W for all of K (8 waves x K/8,
 * summed through LDS in K order): no slabs, no consumer launch.
 *   swl_gemm_rows_supported: 1 when (M, N, K) is a shape the kernel takes.
 *   swl_gemm_rows_add: residual[t, :] += round(x[t, :] . W^T) (stored rounded to the storage dtype, as the reference stores
 *     it) and nothing else. The norm weight and the sums of squares are left to the consumer, which sees every row anyway:
 *   swl_gemm_skinny_packed_partial_nf ("norm on the fly"): swl_gemm_skinny_packed_partial on x = round(r * norm_w) built
 *     while the raw rows r[M, K] are staged, + ssq_out[k_splits][M] = sum r^2 per K-chunk — the row_ssq of
 *     swl_paged_attn_decode_qkv_rs (ssq_parts = k_splits). Even splits only. The fused qkv projection, :46-56.
 *   swl_gemm_skinny_packed_silu_gate_nf: swl_gemm_skinny_packed_silu_gate_rs with rstd from its own sums (:120-127).
 * Together: a decode layer of <= 8 sequences in 5 launches (qkv, attention, o, up/gate, down), 6 up to 32 (down_proj stays
 * split-K + one consumer there). */
int swl_gemm_rows_supported(int32_t M, int32_t N, int32_t K);
int swl_gemm_rows_add(void *residual, const void *x, const void *w_packed, int32_t M, int32_t N, int32_t K,
                      int64_t x_row_stride, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_partial_nf(float *slabs, size_t slabs_bytes, float *ssq_out, const void *x, const void *norm_w,
                                      const void *w_packed, int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                      int32_t k_splits, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_silu_gate_nf(void *out, const void *x, const void *norm_w, float eps,
                                        const void *w_up_gate_packed, int32_t M, int32_t I, int32_t K,
                                        int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, swl_stream_t stream);

/* ---- the same path with the reference's rounding points (r06c: float16 AND bfloat16) -----------------------------------------
 * The deferred norm of the `_nf` entries rounds round(r * w) and applies the 1/rms after the projection — bfloat16 only
 * (DESIGN.md section 4.5). Here the norm is EXACT: fused_add_rmsnorm's arithmetic (reference rmsnorm.py:54-64 at
 * layers/transformer_layer.py:46,120: the sum rounded and stored, the sums of squares and the 1/rms in fp32, ONE rounding of
 * r * rstd * w), split over the two launches that are there anyway:
 *   swl_gemm_rows_add_ssq: swl_gemm_rows_add + ssq_out[M][N / 16] (fp32) = per (token, 16-column tile) sums of squares of the
 *     updated residual rows (the workgroup that finishes a tile has its values in registers);
 *   swl_gemm_skinny_packed_silu_gate_nx / swl_gemm_skinny_packed_partial_nx: the consuming projections add the ssq_parts
 *     (= hidden / 16, % 64 == 0) partials of every row in a fixed order before their first tile and stage
 *     round(r * rstd * norm_w); outputs as swl_gemm_skinny_packed_silu_gate / swl_gemm_skinny_packed_partial (nothing
 *     pending: the slabs go to swl_paged_attn_decode_qkv as they are). */
int swl_gemm_rows_add_ssq(void *residual, float *ssq_out, const void *x, const void *w_packed, int32_t M, int32_t N, int32_t K,
                          int64_t x_row_stride, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_silu_gate_nx(void *out, const void *x, const void *norm_w, float eps, const float *ssq_in,
                                        int32_t ssq_parts, const void *w_up_gate_packed, int32_t M, int32_t I, int32_t K,
                                        int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, swl_stream_t stream);
int swl_gemm_skinny_packed_partial_nx(float *slabs, size_t slabs_bytes, const void *x, const void *norm_w, float eps,
                                      const float *ssq_in, int32_t ssq_parts, const void *w_packed, int32_t M, int32_t N,
                                      int32_t K, int64_t x_row_stride, int32_t k_splits, int32_t dtype, swl_stream_t stream);

/* ---- the transformer stack of a one-sequence decode step as ONE persistent launch (csrc/decode_engine.hip) ----------------
 * reference: the layer loop of LlamaModel._forward (swiftllm/worker/model.py:228-249) over
 * LlamaTransformerLayer.forward (swiftllm/worker/layers/transformer_layer.py:31-130) for ONE decoding sequence: embedding row
 * (pre_layer.py:15-20), and per layer fused add + RMSNorm (rmsnorm.py:39-89), q/k/v projections (linear.py:3-12), rotary
 * (rotary_emb.py:7-58), KV store of the new token (kvcache_mgmt.py:50-79), paged attention phase 1 + 2
 * (paged_attn.py:9-149), o projection, fused add + RMSNorm, up/gate projection, SiLU-gate (silu_and_mul.py:5-34), down
 * projection — with the reference's rounding points in BOTH dtypes. 256 workgroups (one per CU): a loader wave streams a
 * per-CU weight stream by LDS-DMA through a 7 x 16 KiB ring, three consumer waves do the arithmetic; operator boundaries
 * are all-to-all hand-offs of 8-byte {tag, payload} granules inside the launch; every wait is bounded.
 *   swl_decode_engine_supported: 1 when the model shape and the device (num_cus == 256, head_dim == 128, H*D == hidden,
 *     H/KVH in {1, 2, 4}, hidden and ffn_inter_dim multiples of 2048 ...) are what the kernel is laid out for.
 *   swl_decode_engine_slots_per_layer: 16 KiB slots one CU streams per layer (0: unsupported). The weight stream is
 *     [num_layers][256][slots][8192 elements]: per CU and layer qkv | o | up,gate | down; an operator's slots are ordered
 *     (row group of 8, k-chunk of 1024); inside a slot piece p (1 KiB), lane l holds W[row 8g + l/8][1024j + 64p + 8(l%8) ..+8]
 *     (swiftllm_amd/worker/decode_engine.py: pack_engine_layer). CU c owns rows [c*N/256, (c+1)*N/256) of every projection
 *     (up rows and the matching gate rows for the FFN).
 *   swl_decode_engine_workspace_bytes / swl_decode_engine_reset: the hand-off workspace (state words + granule regions);
 *     reset zeroes it and starts the epoch counter (synchronises the stream). Call once after allocation and after any
 *     step that reported an error.
 *   swl_decode_engine_step: resid_out[hidden] = the residual stream after the last layer for the sequence described by
 *     input_ids[0] / seq_ids[0] / seq_lens[0] (length INCLUDING the token being decoded, model.py:296); writes the new
 *     token's K/V into the pools. err_out (may be NULL) receives 0, or the code of a timed-out wait (the workspace then stays
 *     poisoned until reset: later steps return at once with the same code). debug_stamps (may be NULL): [7][num_layers][16]
 *     100 MHz timestamps of the phases of CUs 0, 37, ... 222. flags: bit 0 = the loader keeps one fill in flight instead of
 *     three to four while a consumer wave of its CU sweeps granules (pass 1 unless measuring). Must not run concurrently
 *     with another kernel on the device (one workgroup per CU, all 256 resident). */
int swl_decode_engine_supported(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                int32_t ffn_inter_dim, int32_t num_cus);
int swl_decode_engine_slots_per_layer(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads, int32_t ffn_inter_dim);
size_t swl_decode_engine_workspace_bytes(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads,
                                         int32_t ffn_inter_dim);
int swl_decode_engine_reset(void *workspace, size_t workspace_bytes, swl_stream_t stream);
int swl_decode_engine_step(void *resid_out, const void *w_stream, const void *norms, const void *wte, void *k_cache,
                           void *v_cache, const int32_t *block_table, const int32_t *input_ids, const int32_t *seq_ids,
                           const int32_t *seq_lens, const void *cos_table, const void *sin_table, void *workspace,
                           size_t workspace_bytes, int64_t *err_out, uint64_t *debug_stamps, int32_t num_layers,
                           int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                           int32_t ffn_inter_dim, int32_t max_blocks_per_seq, float eps, float softmax_scale,
                           int32_t flags, int32_t dtype, swl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIFTLLM_HIP_H */
