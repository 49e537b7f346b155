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

// Prefill fusion (r05): rotary on q and k (rotary_emb.py:7-42) AND the prefill KV store
// (kvcache_mgmt.py:10-48) in one pass — r01-r04 walked k twice (rotary_kernel: read + write, then store_kv_prefill_kernel:
// read again) in two launches. grid = (ceil(max_prefill_len / bs), num_prefill_seqs): the workgroup that owns one logical
// block of one sequence rotates its <= 16 tokens' q heads in place, rotates their k heads and writes them BOTH back to k
// (the prefill attention reads the fresh projections, not the pool) and into the contiguous [KVH, bs, D] pool tile, and
// copies v into its tile. Same arithmetic and rounding points as rotary.hip (rotate8), bit-identical stores.
template <typename T>
__global__ __launch_bounds__(256) void rotary_store_prefill_kernel(
    T *__restrict__ q, T *__restrict__ k, const T *__restrict__ v, const T *__restrict__ cos_t, const T *__restrict__ sin_t,
    const int *__restrict__ pos_idx, T *__restrict__ k_cache, T *__restrict__ v_cache, const int *__restrict__ block_table,
    const int *__restrict__ seq_ids, const int *__restrict__ start_locs, const int *__restrict__ seq_lens, int cur_layer,
    int num_layers, int H, int KVH, int block_size, int D, int max_blocks_per_seq, int64_t q_tok_stride, int64_t k_tok_stride,
    int64_t v_tok_stride) {
    const int s = blockIdx.y;
    const int lb = blockIdx.x;
    const int len = seq_lens[s];
    const int tok0 = lb * block_size;
    if (tok0 >= len) return;
    const int ntok = min(block_size, len - tok0);
    const int64_t start = start_locs[s];
    const int seq_id = seq_ids[s];
    const int64_t blk = block_table[static_cast<int64_t>(seq_id) * max_blocks_per_seq + lb];
    const int64_t tile = (blk * num_layers + cur_layer) * KVH * static_cast<int64_t>(block_size) * D;
    const int half = D >> 1;
    const int chunks = D >> 4;                      // 8-element chunks in half a head
    // ---- k: rotate, write back, write into the pool tile; items ordered [kvh][t][chunk] ----
    const int k_items = KVH * block_size * chunks;
    for (int it = threadIdx.x; it < k_items; it += 256) {
        const int c = it % chunks;
        const int t = (it / chunks) % block_size;
        const int h = it / (chunks * block_size);
        if (t >= ntok) continue;
        const int64_t tok = start + tok0 + t;
        const int64_t row = pos_idx ? pos_idx[tok] : tok0 + t;
        const vec8_t<T> cv = load8(cos_t + row * half + c * 8);
        const vec8_t<T> sv = load8(sin_t + row * half + c * 8);
        T *src = k + tok * k_tok_stride + static_cast<int64_t>(h) * D;
        vec8_t<T> x0 = load8(src + c * 8), x1 = load8(src + half + c * 8);
        rotate8<T>(x0, x1, cv, sv);
        store8(src + c * 8, x0);
        store8(src + half + c * 8, x1);
        T *dst = k_cache + tile + (static_cast<int64_t>(h) * block_size + t) * D;
        store8(dst + c * 8, x0);
        store8(dst + half + c * 8, x1);
    }
    // ---- v: copy into the pool tile ----
    const int cpr = D >> 3;
    const int v_items = KVH * block_size * cpr;
    for (int it = threadIdx.x; it < v_items; it += 256) {
        const int c = it % cpr;
        const int t = (it / cpr) % block_size;
        const int h = it / (cpr * block_size);
        if (t < ntok)
            store8(v_cache + tile + static_cast<int64_t>(it) * 8,
                   load8(v + (start + tok0 + t) * v_tok_stride + static_cast<int64_t>(h) * D + c * 8));
    }
    // ---- q: rotate in place; items ordered [t][head][chunk] ----
    const int q_items = ntok * H * chunks;
    for (int it = threadIdx.x; it < q_items; it += 256) {
        const int c = it % chunks;
        const int h = (it / chunks) % H;
        const int t = it / (chunks * H);
        const int64_t tok = start + tok0 + t;
        const int64_t row = pos_idx ? pos_idx[tok] : tok0 + t;
        const vec8_t<T> cv = load8(cos_t + row * half + c * 8);
        const vec8_t<T> sv = load8(sin_t + row * half + c * 8);
        T *src = q + tok * q_tok_stride + static_cast<int64_t>(h) * D;
        vec8_t<T> x0 = load8(src + c * 8), x1 = load8(src + half + c * 8);
        rotate8<T>(x0, x1, cv, sv);
        store8(src + c * 8, x0);
        store8(src + half + c * 8, x1);
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

extern "C" int swl_rotary_store_kv_prefill(void *q, void *k, const void *v, const void *cos_table, const void *sin_table,
                                           const int32_t *pos_idx, void *k_cache, void *v_cache, const int32_t *block_table,
                                           const int32_t *seq_ids, const int32_t *start_locs, const int32_t *seq_lens,
                                           int32_t num_prefill_seqs, int32_t max_prefill_len, int32_t cur_layer,
                                           int32_t num_layers, int32_t num_q_heads, int32_t num_kv_heads, int32_t block_size,
                                           int32_t head_dim, int32_t max_blocks_per_seq, int64_t q_tok_stride,
                                           int64_t k_tok_stride, int64_t v_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_prefill_seqs < 0 || max_prefill_len < 0) return SWL_ERR_BAD_ARG;
    if (num_prefill_seqs == 0 || max_prefill_len == 0) return SWL_OK;
    if (!store_args_ok(k_cache, v_cache, k, v, block_table, seq_ids, seq_lens, cur_layer, num_layers, num_kv_heads,
                       block_size, head_dim, max_blocks_per_seq, k_tok_stride, v_tok_stride) ||
        !start_locs || !q || !cos_table || !sin_table || num_q_heads <= 0)
        return SWL_ERR_BAD_ARG;
    if (!(head_dim == 32 || head_dim == 64 || head_dim == 128 || head_dim == 256)) return SWL_ERR_UNSUPPORTED;
    if ((q_tok_stride & 7) || q_tok_stride < static_cast<int64_t>(num_q_heads) * head_dim || !swl::aligned16(q) ||
        !swl::aligned16(cos_table) || !swl::aligned16(sin_table))
        return SWL_ERR_BAD_ARG;
    if (num_prefill_seqs > 65535) return SWL_ERR_UNSUPPORTED;
    const dim3 grid((max_prefill_len + block_size - 1) / block_size, num_prefill_seqs);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::rotary_store_prefill_kernel<T>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<T *>(q), static_cast<T *>(k), static_cast<const T *>(v),
                           static_cast<const T *>(cos_table), static_cast<const T *>(sin_table), pos_idx,
                           static_cast<T *>(k_cache), static_cast<T *>(v_cache), block_table, seq_ids, start_locs, seq_lens,
                           cur_layer, num_layers, num_q_heads, num_kv_heads, block_size, head_dim, max_blocks_per_seq,
                           q_tok_stride, k_tok_stride, v_tok_stride);
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
