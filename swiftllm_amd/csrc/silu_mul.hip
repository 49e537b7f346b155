// silu_mul.hip — SiLU-gate of the LLaMA FFN, in place, for gfx950.
//
// Replaces _fwd_silu_and_mul (swiftllm/worker/kernels/silu_and_mul.py:5-23):
//   x[:, :I] <- x[:, :I] * round_T( g / (1 + exp(-g)) ),  g = fp32(x[:, I:])
// (up_gate_proj rows are [up ; gate], weight.py:133). HBM-bound: 3*T*I*e bytes.
// Rounding points as the reference: silu in fp32, rounded to the storage dtype, then the product
// up*gate rounded to the storage dtype (silu_and_mul.py:18-22).
// Mapping: grid-stride over 16-byte chunks; both the up and the gate chunk of a lane are contiguous
// 16-byte accesses, so a wave moves 1 KiB per instruction.
#include "swl_common.h"

namespace swl {

template <typename T>
__global__ __launch_bounds__(256) void silu_mul_kernel(T *__restrict__ x, int64_t num_items,
                                                       int chunks_per_row, int I) {
    for (int64_t item = blockIdx.x * 256ll + threadIdx.x; item < num_items;
         item += static_cast<int64_t>(gridDim.x) * 256ll) {
        const int64_t tok = item / chunks_per_row;
        const int c = static_cast<int>(item - tok * chunks_per_row);
        T *up_p = x + tok * (2ll * I) + c * 8;
        vec8_t<T> up = load8(up_p);
        const vec8_t<T> gate = load8_nt(up_p + I); // gate half is dead after this kernel
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = to_f(gate[j]);
            const T act = to_t<T>(g / (1.0f + expf(-g)));
            up[j] = mul_t<T>(up[j], act);
        }
        store8(up_p, up);
    }
}

} // namespace swl

extern "C" int swl_silu_mul(void *x, int64_t num_tokens, int32_t ffn_inter_dim, int32_t dtype,
                            swl_stream_t stream) {
    if (num_tokens < 0 || ffn_inter_dim <= 0 || (ffn_inter_dim & 7)) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (!x || !swl::aligned16(x)) return SWL_ERR_BAD_ARG;
    const int chunks = ffn_inter_dim / 8;
    const int64_t items = num_tokens * chunks;
    const int64_t blocks = (items + 255) / 256;
    // <= 8 workgroups per CU x 256 CUs, grid-stride the rest
    const unsigned grid = static_cast<unsigned>(blocks < 16384 ? blocks : 16384);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::silu_mul_kernel<T>), dim3(grid), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(x), items, chunks,
                           ffn_inter_dim);
    });
    return swl::check_launch();
}
