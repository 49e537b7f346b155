// block_table.hip — device-resident block-table maintenance for the paged KV pools (gfx950).
//
// Replaces the three Triton kernels of swiftllm/worker/kernels/block_mgmt.py:
//   _fwd_set_block_table_and_num_seq_alloc_blocks_kernel   (block_mgmt.py:5-24)
//   _fwd_unset_block_table_and_num_seq_alloc_blocks_kernel (block_mgmt.py:49-64)
//   _fwd_gather_allocated_blocks_and_unset_kernel          (block_mgmt.py:83-104)
// Integer work, bit-exact. One wave per batch entry; the lanes stride over the sequence's blocks
// (the reference walks them with one scalar program), so a 1024-block prefill row is 16 coalesced
// 256-byte stores instead of 1024 dependent scalar stores.
#include "swl_common.h"

namespace swl {

__global__ __launch_bounds__(64) void block_table_set_kernel(
    int *__restrict__ num_alloc, int *__restrict__ block_table, const int *__restrict__ candidates,
    const int *__restrict__ seq_ids, const int *__restrict__ need, const int *__restrict__ need_off,
    uint8_t *__restrict__ is_block_free, int max_blocks_per_seq) {
    const int i = blockIdx.x;
    const int64_t s = seq_ids[i];
    const int n = need[i];
    const int off = need_off[i];
    const int have = num_alloc[s];
    int *row = block_table + s * max_blocks_per_seq + have;
    for (int j = threadIdx.x; j < n; j += 64) {
        const int b = candidates[off + j];
        row[j] = b;
        if (is_block_free) is_block_free[b] = 0;
    }
    __syncthreads(); // every lane has read `have` before lane 0 overwrites it
    if (threadIdx.x == 0) num_alloc[s] = have + n;
}

template <bool GATHER>
__global__ __launch_bounds__(64) void block_table_unset_kernel(
    int *__restrict__ num_alloc, const int *__restrict__ block_table,
    const int *__restrict__ seq_ids, uint8_t *__restrict__ is_block_free,
    const int *__restrict__ out_off, int *__restrict__ gathered, int max_blocks_per_seq) {
    const int i = blockIdx.x;
    const int64_t s = seq_ids[i];
    const int n = num_alloc[s];
    const int *row = block_table + s * max_blocks_per_seq;
    const int off = GATHER ? out_off[i] : 0;
    for (int j = threadIdx.x; j < n; j += 64) {
        const int b = row[j];
        if (GATHER) gathered[off + j] = b;
        is_block_free[b] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) num_alloc[s] = 0;
}

} // namespace swl

extern "C" int swl_block_table_set(int32_t *num_seq_allocated_blocks, int32_t *block_table,
                                   const int32_t *candidate_blocks, const int32_t *seq_ids,
                                   const int32_t *block_needed,
                                   const int32_t *block_needed_excl_cumsum,
                                   uint8_t *is_block_free, int32_t batch_size,
                                   int32_t max_blocks_per_seq, swl_stream_t stream) {
    if (batch_size < 0 || max_blocks_per_seq <= 0) return SWL_ERR_BAD_ARG;
    if (batch_size == 0) return SWL_OK;
    if (!num_seq_allocated_blocks || !block_table || !seq_ids || !block_needed ||
        !block_needed_excl_cumsum)
        return SWL_ERR_BAD_ARG; // candidate_blocks may be NULL when nothing is needed
    hipLaunchKernelGGL(swl::block_table_set_kernel, dim3(batch_size), dim3(64), 0,
                       static_cast<hipStream_t>(stream), num_seq_allocated_blocks, block_table,
                       candidate_blocks, seq_ids, block_needed, block_needed_excl_cumsum,
                       is_block_free, max_blocks_per_seq);
    return swl::check_launch();
}

extern "C" int swl_block_table_unset(int32_t *num_seq_allocated_blocks, const int32_t *block_table,
                                     const int32_t *seq_ids, uint8_t *is_block_free,
                                     int32_t batch_size, int32_t max_blocks_per_seq,
                                     swl_stream_t stream) {
    if (batch_size < 0 || max_blocks_per_seq <= 0) return SWL_ERR_BAD_ARG;
    if (batch_size == 0) return SWL_OK;
    if (!num_seq_allocated_blocks || !block_table || !seq_ids || !is_block_free)
        return SWL_ERR_BAD_ARG;
    hipLaunchKernelGGL(swl::block_table_unset_kernel<false>, dim3(batch_size), dim3(64), 0,
                       static_cast<hipStream_t>(stream), num_seq_allocated_blocks, block_table,
                       seq_ids, is_block_free, nullptr, nullptr, max_blocks_per_seq);
    return swl::check_launch();
}

extern "C" int swl_block_table_gather(int32_t *num_seq_allocated_blocks,
                                      const int32_t *block_table, const int32_t *seq_ids,
                                      uint8_t *is_block_free, const int32_t *out_excl_cumsum,
                                      int32_t *gathered_block_ids, int32_t batch_size,
                                      int32_t max_blocks_per_seq, swl_stream_t stream) {
    if (batch_size < 0 || max_blocks_per_seq <= 0) return SWL_ERR_BAD_ARG;
    if (batch_size == 0) return SWL_OK;
    if (!num_seq_allocated_blocks || !block_table || !seq_ids || !is_block_free ||
        !out_excl_cumsum)
        return SWL_ERR_BAD_ARG; // gathered_block_ids may be NULL when every sequence is empty
    hipLaunchKernelGGL(swl::block_table_unset_kernel<true>, dim3(batch_size), dim3(64), 0,
                       static_cast<hipStream_t>(stream), num_seq_allocated_blocks, block_table,
                       seq_ids, is_block_free, out_excl_cumsum, gathered_block_ids,
                       max_blocks_per_seq);
    return swl::check_launch();
}
