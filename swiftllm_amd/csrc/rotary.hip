// rotary.hip — rotate-half rotary embedding (in place) for gfx950, plus the decode-path fusion
// rotary + KV-store.
//
// Replaces _fwd_rotary_embedding (swiftllm/worker/kernels/rotary_emb.py:7-42). HBM-bound:
// 2*T*(H+KVH)*D*e bytes + the cos/sin rows (L2 resident across heads).
// Mapping: one lane owns 8 consecutive elements of the first half of a head and the matching 8 of
// the second half (two 16-byte loads, two 16-byte stores); cos/sin come as one 16-byte load each.
// Rounding points: the reference evaluates q0*cos - q1*sin and q0*sin + q1*cos in the storage
// dtype (rotary_emb.py:34-42), i.e. every product and every sum is rounded. We do the same
// (mul_t / add_t / sub_t round once per operation, no fma contraction).
#include "swl_common.h"

namespace swl {

// Work item = (token, head in [0, H+KVH), chunk in [0, D/16)).
template <typename T>
__global__ __launch_bounds__(256) void rotary_kernel(T *__restrict__ q, T *__restrict__ k,
                                                     const T *__restrict__ cos_t,
                                                     const T *__restrict__ sin_t,
                                                     const int *__restrict__ pos_idx,
                                                     int64_t num_items, int H, int KVH, int D,
                                                     int64_t q_tok_stride, int64_t k_tok_stride) {
    const int chunks = D >> 4; // 8-element chunks in half a head
    const int heads = H + KVH;
    for (int64_t item = blockIdx.x * 256ll + threadIdx.x; item < num_items;
         item += static_cast<int64_t>(gridDim.x) * 256ll) {
        const int c = static_cast<int>(item % chunks);
        const int64_t th = item / chunks;
        const int hh = static_cast<int>(th % heads);
        const int64_t tok = th / heads;
        const int64_t row = pos_idx ? pos_idx[tok] : tok;
        if (row < 0) continue; // an inert row of a padded decode batch (length 0: worker/model.py, hipGraph buckets)
        const vec8_t<T> cv = load8(cos_t + row * (D >> 1) + c * 8);
        const vec8_t<T> sv = load8(sin_t + row * (D >> 1) + c * 8);
        T *base = hh < H ? q + tok * q_tok_stride + static_cast<int64_t>(hh) * D
                         : k + tok * k_tok_stride + static_cast<int64_t>(hh - H) * D;
        vec8_t<T> x0 = load8(base + c * 8);
        vec8_t<T> x1 = load8(base + (D >> 1) + c * 8);
        rotate8<T>(x0, x1, cv, sv);
        store8(base + c * 8, x0);
        store8(base + (D >> 1) + c * 8, x1);
    }
}

// Decode fusion: one workgroup per decoding sequence. Rotates the token's q and k heads in place and
// writes the rotated k and the v row straight into the paged pools
// (= kvcache_mgmt.py:50-79 applied after rotary_emb.py). Items [0, (H+KVH)*D/16) rotate,
// items after that copy v (D/8 chunks per kv head).
template <typename T>
__global__ __launch_bounds__(256) void rotary_store_decode_kernel(
    T *__restrict__ q, T *__restrict__ k, const T *__restrict__ v, const T *__restrict__ cos_t,
    const T *__restrict__ sin_t, const int *__restrict__ pos_idx, T *__restrict__ k_cache,
    T *__restrict__ v_cache, const int *__restrict__ block_table, const int *__restrict__ seq_ids,
    const int *__restrict__ seq_lens, int H, int KVH, int D, int cur_layer, int num_layers,
    int block_size, int max_blocks_per_seq, int64_t q_tok_stride, int64_t k_tok_stride,
    int64_t v_tok_stride, const float *__restrict__ slabs, int ks, int64_t slab_stride) {
    // With `slabs` the inputs are the split-K partials of the fused qkv projection, [ks][tokens]
    // [(H + 2*KVH)*D] fp32: they are summed and rounded here, and q/k/v are pure OUTPUT buffers.
    const int64_t tok = blockIdx.x;
    const int64_t qkv_row = static_cast<int64_t>(H + 2 * KVH) * D;
    const int seq_id = seq_ids[tok];
    const int pos = seq_lens[tok] - 1;
    if (pos < 0) return; // an inert row of a padded decode batch (length 0): nothing rotated, nothing stored
    const int64_t row = pos_idx ? pos_idx[tok] : pos;
    const int64_t blk = block_table[static_cast<int64_t>(seq_id) * max_blocks_per_seq + pos / block_size];
    const int slot = pos % block_size;
    // pool offset of (blk, layer, kvh=0, slot, 0)
    const int64_t pool_base =
        ((blk * num_layers + cur_layer) * KVH) * static_cast<int64_t>(block_size) * D +
        static_cast<int64_t>(slot) * D;
    const int64_t head_pitch = static_cast<int64_t>(block_size) * D;

    const int chunks = D >> 4;
    const int rot_items = (H + KVH) * chunks;
    const int v_items = KVH * (D >> 3);
    // one item per thread: grid.y workgroups share a token, so the (latency-bound) load -> rotate ->
    // store chain is walked once, not once per 256 items
    for (int item = blockIdx.y * 256 + threadIdx.x; item < rot_items + v_items; item += gridDim.y * 256) {
        if (item < rot_items) {
            const int c = item % chunks;
            const int hh = item / chunks;
            const vec8_t<T> cv = load8(cos_t + row * (D >> 1) + c * 8);
            const vec8_t<T> sv = load8(sin_t + row * (D >> 1) + c * 8);
            const bool is_q = hh < H;
            T *base = is_q ? q + tok * q_tok_stride + static_cast<int64_t>(hh) * D
                           : k + tok * k_tok_stride + static_cast<int64_t>(hh - H) * D;
            vec8_t<T> x0, x1;
            if (slabs) { // q heads then k heads are contiguous in a fused-qkv row: offset hh*D for both
                const int64_t off = tok * qkv_row + static_cast<int64_t>(hh) * D + c * 8;
                x0 = load8_splitk<T>(slabs, ks, slab_stride, off);
                x1 = load8_splitk<T>(slabs, ks, slab_stride, off + (D >> 1));
            } else {
                x0 = load8(base + c * 8);
                x1 = load8(base + (D >> 1) + c * 8);
            }
            rotate8<T>(x0, x1, cv, sv);
            store8(base + c * 8, x0);
            store8(base + (D >> 1) + c * 8, x1);
            if (!is_q) {
                T *dst = k_cache + pool_base + (hh - H) * head_pitch;
                store8(dst + c * 8, x0);
                store8(dst + (D >> 1) + c * 8, x1);
            }
        } else {
            const int vi = item - rot_items;
            const int c = vi % (D >> 3);
            const int kvh = vi / (D >> 3);
            vec8_t<T> vv;
            if (slabs) {
                vv = load8_splitk<T>(slabs, ks, slab_stride,
                                     tok * qkv_row + static_cast<int64_t>(H + KVH + kvh) * D + c * 8);
                store8(const_cast<T *>(v) + tok * v_tok_stride + static_cast<int64_t>(kvh) * D + c * 8, vv);
            } else {
                vv = load8(v + tok * v_tok_stride + static_cast<int64_t>(kvh) * D + c * 8);
            }
            store8(v_cache + pool_base + kvh * head_pitch + c * 8, vv);
        }
    }
}

__global__ void decode_positions_kernel(int *__restrict__ pos_idx, const int *__restrict__ seq_lens,
                                        int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos_idx[i] = seq_lens[i] - 1;
}

} // namespace swl

static bool rotary_args_ok(const void *q, const void *k, const void *c, const void *s, int H,
                           int KVH, int D, int64_t qs, int64_t ks) {
    if (!q || !k || !c || !s) return false;
    if (H <= 0 || KVH <= 0) return false;
    if (!(D == 32 || D == 64 || D == 128 || D == 256)) return false;
    if (qs < static_cast<int64_t>(H) * D || ks < static_cast<int64_t>(KVH) * D) return false;
    if ((qs & 7) || (ks & 7)) return false;
    return swl::aligned16(q) && swl::aligned16(k) && swl::aligned16(c) && swl::aligned16(s);
}

extern "C" int swl_rotary(void *q, void *k, const void *cos_table, const void *sin_table,
                          const int32_t *pos_idx, int64_t num_tokens, int32_t num_q_heads,
                          int32_t num_kv_heads, int32_t head_dim, int64_t q_tok_stride,
                          int64_t k_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_tokens < 0) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (!rotary_args_ok(q, k, cos_table, sin_table, num_q_heads, num_kv_heads, head_dim,
                        q_tok_stride, k_tok_stride))
        return SWL_ERR_BAD_ARG;
    const int64_t items = num_tokens * (num_q_heads + num_kv_heads) * (head_dim / 16);
    const int64_t blocks = (items + 255) / 256;
    const unsigned grid = static_cast<unsigned>(blocks < 65536 ? blocks : 65536);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::rotary_kernel<T>), dim3(grid), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(q),
                           static_cast<T *>(k), static_cast<const T *>(cos_table),
                           static_cast<const T *>(sin_table), pos_idx, items, num_q_heads,
                           num_kv_heads, head_dim, q_tok_stride, k_tok_stride);
    });
    return swl::check_launch();
}

static int rotary_store_decode_impl(
    void *q, void *k, void *v, const void *cos_table, const void *sin_table, const int32_t *pos_idx,
    void *k_cache, void *v_cache, const int32_t *block_table, const int32_t *seq_ids,
    const int32_t *seq_lens, int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads,
    int32_t head_dim, int32_t cur_layer, int32_t num_layers, int32_t block_size,
    int32_t max_blocks_per_seq, int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
    const float *slabs, int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0) return SWL_OK;
    if (!rotary_args_ok(q, k, cos_table, sin_table, num_q_heads, num_kv_heads, head_dim,
                        q_tok_stride, k_tok_stride))
        return SWL_ERR_BAD_ARG;
    if (!v || !k_cache || !v_cache || !block_table || !seq_ids || !seq_lens) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(v) || !swl::aligned16(k_cache) || !swl::aligned16(v_cache) ||
        (v_tok_stride & 7) || v_tok_stride < static_cast<int64_t>(num_kv_heads) * head_dim)
        return SWL_ERR_BAD_ARG;
    if (block_size <= 0 || num_layers <= 0 || cur_layer < 0 || cur_layer >= num_layers ||
        max_blocks_per_seq <= 0)
        return SWL_ERR_BAD_ARG;
    if (slabs && (k_splits <= 0 || !swl::aligned16(slabs))) return SWL_ERR_BAD_ARG;
    const int64_t slab_stride = static_cast<int64_t>(num_decoding_seqs) *
                                (num_q_heads + 2 * num_kv_heads) * head_dim;
    SWL_DISPATCH_DTYPE(dtype, T, {
        const int items = (num_q_heads + num_kv_heads) * (head_dim / 16) + num_kv_heads * (head_dim / 8);
        const int wgs_per_token = items > 2048 ? 8 : (items + 255) / 256;
        hipLaunchKernelGGL((swl::rotary_store_decode_kernel<T>), dim3(num_decoding_seqs, wgs_per_token), dim3(256),
                           0, static_cast<hipStream_t>(stream), static_cast<T *>(q),
                           static_cast<T *>(k), static_cast<const T *>(v),
                           static_cast<const T *>(cos_table), static_cast<const T *>(sin_table),
                           pos_idx, static_cast<T *>(k_cache), static_cast<T *>(v_cache),
                           block_table, seq_ids, seq_lens, num_q_heads, num_kv_heads, head_dim,
                           cur_layer, num_layers, block_size, max_blocks_per_seq, q_tok_stride,
                           k_tok_stride, v_tok_stride, slabs, k_splits, slab_stride);
    });
    return swl::check_launch();
}

extern "C" int swl_rotary_store_kv_decode(
    void *q, void *k, const void *v, const void *cos_table, const void *sin_table,
    const int32_t *pos_idx, void *k_cache, void *v_cache, const int32_t *block_table,
    const int32_t *seq_ids, const int32_t *seq_lens, int32_t num_decoding_seqs, int32_t num_q_heads,
    int32_t num_kv_heads, int32_t head_dim, int32_t cur_layer, int32_t num_layers,
    int32_t block_size, int32_t max_blocks_per_seq, int64_t q_tok_stride, int64_t k_tok_stride,
    int64_t v_tok_stride, int32_t dtype, swl_stream_t stream) {
    return rotary_store_decode_impl(q, k, const_cast<void *>(v), cos_table, sin_table, pos_idx, k_cache,
                                    v_cache, block_table, seq_ids, seq_lens, num_decoding_seqs,
                                    num_q_heads, num_kv_heads, head_dim, cur_layer, num_layers,
                                    block_size, max_blocks_per_seq, q_tok_stride, k_tok_stride,
                                    v_tok_stride, nullptr, 0, dtype, stream);
}

extern "C" int swl_splitk_rotary_store_kv_decode(
    void *q_out, void *k_out, void *v_out, const float *qkv_slabs, int32_t k_splits,
    const void *cos_table, const void *sin_table, const int32_t *pos_idx, void *k_cache,
    void *v_cache, const int32_t *block_table, const int32_t *seq_ids, const int32_t *seq_lens,
    int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
    int32_t cur_layer, int32_t num_layers, int32_t block_size, int32_t max_blocks_per_seq,
    int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride, int32_t dtype,
    swl_stream_t stream) {
    if (!qkv_slabs) return SWL_ERR_BAD_ARG;
    return rotary_store_decode_impl(q_out, k_out, v_out, cos_table, sin_table, pos_idx, k_cache, v_cache,
                                    block_table, seq_ids, seq_lens, num_decoding_seqs, num_q_heads,
                                    num_kv_heads, head_dim, cur_layer, num_layers, block_size,
                                    max_blocks_per_seq, q_tok_stride, k_tok_stride, v_tok_stride,
                                    qkv_slabs, k_splits, dtype, stream);
}

extern "C" int swl_decode_positions(int32_t *pos_idx, const int32_t *seq_lens,
                                    int32_t num_decoding_seqs, swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0) return SWL_OK;
    if (!pos_idx || !seq_lens) return SWL_ERR_BAD_ARG;
    hipLaunchKernelGGL(swl::decode_positions_kernel, dim3((num_decoding_seqs + 255) / 256),
                       dim3(256), 0, static_cast<hipStream_t>(stream), pos_idx, seq_lens,
                       num_decoding_seqs);
    return swl::check_launch();
}
