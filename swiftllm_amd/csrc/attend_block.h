// attend_block.h — the VALU form of one 16-token KV block of decode attention, shared by paged_attn.hip (the G = 1
// path of the flash-decoding kernel) and decode_engine.hip (the batch-1 persistent decode step).
// Reference: swiftllm/worker/kernels/paged_attn.py:45-108 (the online-softmax loop of phase 1).
#pragma once

#include "swl_common.h"

namespace swl {

constexpr int kBlk = 16;          // tokens per KV block (engine_config.block_size)

template <typename T, int D, int G>
struct DecodeTile {
    static constexpr int LPT = D / 8;      // lanes per token row
    static constexpr int TPI = 64 / LPT;   // tokens per load instruction (rows per wave)
    static constexpr int NI = kBlk / TPI;  // load instructions per 16-token block
};

// One 16-token block for one wave. s/p live only here; m, l, acc persist. `partial` (wave-uniform): the block
// crosses the end of the sequence and its tail tokens are masked out; the two selects sit behind a uniform branch
// so the kernel carries ONE copy of this body per ring slot instead of a masked and an unmasked one.
template <typename T, int D, int G>
__device__ __forceinline__ void attend_block(const vec8_t<T> (&qv)[G],
                                             const vec8_t<T> (&Kv)[DecodeTile<T, D, G>::NI],
                                             const vec8_t<T> (&Vv)[DecodeTile<T, D, G>::NI],
                                             float (&m)[G], float (&l)[G], float (&acc)[G][8],
                                             float c, int tok0, int row, int len, bool partial) {
    using Tile = DecodeTile<T, D, G>;
    constexpr int NI = Tile::NI;
    float vf[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[i][j] = to_f(Vv[i][j]);

    bool valid[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) valid[i] = tok0 + i * Tile::TPI + row < len;

#pragma unroll
    for (int g = 0; g < G; ++g) {
        float s[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) s[i] = group_allreduce_sum<Tile::LPT>(dot8<T>(qv[g], Kv[i], 0.f));
        if (partial) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (!valid[i]) s[i] = kNegBig;
        }
        float m_new = m[g];
#pragma unroll
        for (int i = 0; i < NI; ++i) m_new = fmaxf(m_new, s[i]);
        const float mc = m_new * c;
        // difference FIRST: with both maxima at the -1e30 sentinel (a row that has seen no valid
        // token yet) fma(m, c, -mc) would return the rounding residual of the product (~1e22) and
        // exp2 of that is inf; (m - m_new) is exactly 0.
        const float alpha = fast_exp2((m[g] - m_new) * c);
        float p[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) p[i] = fast_exp2(fmaf(s[i], c, -mc));
        if (partial) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (!valid[i]) p[i] = 0.f;
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) psum += p[i];
        l[g] = fmaf(l[g], alpha, psum);
        m[g] = m_new;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = acc[g][j] * alpha;
#pragma unroll
            for (int i = 0; i < NI; ++i) a = fmaf(p[i], vf[i][j], a);
            acc[g][j] = a;
        }
    }
}

} // namespace swl
