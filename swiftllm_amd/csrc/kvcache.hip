// kvcache.hip — store freshly projected K/V rows into the paged KV pools (gfx950).
//
// Replaces _fwd_kvcache_mgmt_prefill_kernel (swiftllm/worker/kernels/kvcache_mgmt.py:10-48) and
// _fwd_kvcache_mgmt_decoding_kernel (kvcache_mgmt.py:50-79). Bit-exact copies, HBM-bound:
// 4*T*KVH*D*e bytes (K and V, read + write).
// Pool layout [num_blocks, L, KVH, block_size, D]: the [KVH, block_size, D] tile of one
// (block, layer) is contiguous (32 KiB for Llama-3-8B), so a workgroup that owns one logical block
// of one sequence writes one fully contiguous 32 KiB run per pool with 16-byte stores; reads are
// 256-byte rows (one head of one token) strided by the token pitch.
// All pool offsets are int64 (a 288 GB pool has > 2^31 elements).
#include "swl_common.h"

namespace swl {

// grid = (ceil(max_prefill_len / bs), num_prefill_seqs)
template <typename T>
__global__ __launch_bounds__(256) void store_kv_prefill_kernel(
    T *__restrict__ k_cache, T *__restrict__ v_cache, const T *__restrict__ k,
    const T *__restrict__ v, const int *__restrict__ block_table, const int *__restrict__ seq_ids,
    const int *__restrict__ start_locs, const int *__restrict__ seq_lens, int cur_layer,
    int num_layers, int KVH, int block_size, int D, int max_blocks_per_seq, int64_t k_tok_stride,
    int64_t v_tok_stride) {
    const int s = blockIdx.y;
    const int lb = blockIdx.x; // logical block inside the sequence
    const int len = seq_lens[s];
    const int tok0 = lb * block_size;
    if (tok0 >= len) return;
    const int ntok = min(block_size, len - tok0);
    const int64_t start = start_locs[s];
    const int seq_id = seq_ids[s];
    const int64_t blk = block_table[static_cast<int64_t>(seq_id) * max_blocks_per_seq + lb];
    const int64_t tile = (blk * num_layers + cur_layer) * KVH * static_cast<int64_t>(block_size) * D;

    const int cpr = D >> 3;                    // 16-byte chunks per (token, head) row
    const int items = KVH * block_size * cpr;  // destination order: [kvh][t][chunk] == contiguous
    for (int it = threadIdx.x; it < items; it += 256) {
        const int c = it % cpr;
        const int t = (it / cpr) % block_size;
        const int h = it / (cpr * block_size);
        if (t < ntok) {
            const int64_t tok = start + tok0 + t;
            const int64_t dst = tile + static_cast<int64_t>(it) * 8;
            store8(k_cache + dst, load8(k + tok * k_tok_stride + static_cast<int64_t>(h) * D + c * 8));
            store8(v_cache + dst, load8(v + tok * v_tok_stride + static_cast<int64_t>(h) * D + c * 8));
        }
    }
}

// grid = (num_decoding_seqs)
template <typename T>
__global__ __launch_bounds__(128) void store_kv_decode_kernel(
    T *__restrict__ k_cache, T *__restrict__ v_cache, const T *__restrict__ k,
    const T *__restrict__ v, const int *__restrict__ block_table, const int *__restrict__ seq_ids,
    const int *__restrict__ seq_lens, int cur_layer, int num_layers, int KVH, int block_size, int D,
    int max_blocks_per_seq, int64_t k_tok_stride, int64_t v_tok_stride) {
    const int64_t i = blockIdx.x;
    const int seq_id = seq_ids[i];
    const int pos = seq_lens[i] - 1;
    if (pos < 0) return; // an inert row of a padded decode batch (length 0)
    const int64_t blk = block_table[static_cast<int64_t>(seq_id) * max_blocks_per_seq + pos / block_size];
    const int slot = pos % block_size;
    const int64_t base = (blk * num_layers + cur_layer) * KVH * static_cast<int64_t>(block_size) * D +
                         static_cast<int64_t>(slot) * D;
    const int cpr = D >> 3;
    for (int it = threadIdx.x; it < KVH * cpr; it += 128) {
        const int c = it % cpr;
        const int h = it / cpr;
        const int64_t dst = base + static_cast<int64_t>(h) * block_size * D + c * 8;
        store8(k_cache + dst, load8(k + i * k_tok_stride + static_cast<int64_t>(h) * D + c * 8));
        store8(v_cache + dst, load8(v + i * v_tok_stride + static_cast<int64_t>(h) * D + c * 8));
    }
}

} // namespace swl

static bool store_args_ok(const void *kc, const void *vc, const void *k, const void *v,
                          const void *bt, const void *ids, const void *lens, int cur_layer, int L,
                          int KVH, int bs, int D, int mbps, int64_t ks, int64_t vs) {
    if (!kc || !vc || !k || !v || !bt || !ids || !lens) return false;
    if (L <= 0 || cur_layer < 0 || cur_layer >= L || KVH <= 0 || bs <= 0 || D <= 0 || (D & 7) ||
        mbps <= 0)
        return false;
    if (ks < static_cast<int64_t>(KVH) * D || vs < static_cast<int64_t>(KVH) * D || (ks & 7) ||
        (vs & 7))
        return false;
    return swl::aligned16(kc) && swl::aligned16(vc) && swl::aligned16(k) && swl::aligned16(v);
}

extern "C" int swl_store_kv_prefill(void *k_cache, void *v_cache, const void *k, const void *v,
                                    const int32_t *block_table, const int32_t *seq_ids,
                                    const int32_t *start_locs, const int32_t *seq_lens,
                                    int32_t num_prefill_seqs, int32_t max_prefill_len,
                                    int32_t cur_layer, int32_t num_layers, int32_t num_kv_heads,
                                    int32_t block_size, int32_t head_dim,
                                    int32_t max_blocks_per_seq, int64_t k_tok_stride,
                                    int64_t v_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_prefill_seqs < 0 || max_prefill_len < 0) return SWL_ERR_BAD_ARG;
    if (num_prefill_seqs == 0 || max_prefill_len == 0) return SWL_OK;
    if (!store_args_ok(k_cache, v_cache, k, v, block_table, seq_ids, seq_lens, cur_layer,
                       num_layers, num_kv_heads, block_size, head_dim, max_blocks_per_seq,
                       k_tok_stride, v_tok_stride) ||
        !start_locs)
        return SWL_ERR_BAD_ARG;
    if (num_prefill_seqs > 65535) return SWL_ERR_UNSUPPORTED;
    const dim3 grid((max_prefill_len + block_size - 1) / block_size, num_prefill_seqs);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::store_kv_prefill_kernel<T>), grid, dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(k_cache),
                           static_cast<T *>(v_cache), static_cast<const T *>(k),
                           static_cast<const T *>(v), block_table, seq_ids, start_locs, seq_lens,
                           cur_layer, num_layers, num_kv_heads, block_size, head_dim,
                           max_blocks_per_seq, k_tok_stride, v_tok_stride);
    });
    return swl::check_launch();
}

extern "C" int swl_store_kv_decode(void *k_cache, void *v_cache, const void *k, const void *v,
                                   const int32_t *block_table, const int32_t *seq_ids,
                                   const int32_t *seq_lens, int32_t num_decoding_seqs,
                                   int32_t cur_layer, int32_t num_layers, int32_t num_kv_heads,
                                   int32_t block_size, int32_t head_dim, int32_t max_blocks_per_seq,
                                   int64_t k_tok_stride, int64_t v_tok_stride, int32_t dtype,
                                   swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0) return SWL_OK;
    if (!store_args_ok(k_cache, v_cache, k, v, block_table, seq_ids, seq_lens, cur_layer,
                       num_layers, num_kv_heads, block_size, head_dim, max_blocks_per_seq,
                       k_tok_stride, v_tok_stride))
        return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::store_kv_decode_kernel<T>), dim3(num_decoding_seqs), dim3(128), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(k_cache),
                           static_cast<T *>(v_cache), static_cast<const T *>(k),
                           static_cast<const T *>(v), block_table, seq_ids, seq_lens, cur_layer,
                           num_layers, num_kv_heads, block_size, head_dim, max_blocks_per_seq,
                           k_tok_stride, v_tok_stride);
    });
    return swl::check_launch();
}
