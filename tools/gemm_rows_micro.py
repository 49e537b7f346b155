#!/usr/bin/env python3
"""gemm_rows_micro.py — swl_gemm_rows_add_scale (csrc/gemm_rows.hip: o_proj with K split inside the workgroup, residual add
and the deferred norm's element-wise half in its epilogue) against the pair it replaces (swl_gemm_skinny_packed_partial +
swl_splitk_add_scale), alone and in the chain o_proj -> [consumer] -> up/gate SiLU GEMM -> down, at Llama-3-8B widths (GPU).
Checks both forms against an fp64 product first; times launches that cycle through distinct weight copies, eagerly and as
one captured hipGraph per chain (what the decode step replays). One JSON line per batch size."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker.kernels.linear import (pack_weight, linear_splitk, linear_rows_add_scale, rows_add_scale_ok,
                                                linear_silu_gate, SplitKPartials, linear_rows_add, linear_splitk_nf,
                                                linear_silu_gate_nf, linear_splitk_from_splitk, linear_silu_gate_from_splitk,
                                                alt_residual_like, tiny_from_splitk_ok)
from swiftllm_amd.worker.kernels.rmsnorm import add_scale_from_splitk


def time_us(fn, iters, warm=3):
    for i in range(warm):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def graph_us(body, copies, iters):
    """`body(i)` for i in range(copies) captured as ONE graph (dependent chain on one stream), replayed iters times."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(copies):
            body(i)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(copies):
            body(i)
    return time_us(lambda i: g.replay(), iters) / copies


def layer_mode(a):
    """The projection side of one decode layer (attention replaced by a fixed tensor) three ways, captured as graphs:
    old    add_scale <- down slabs | qkv split-K | o split-K | add_scale | SiLU-gate (rs) | down split-K        (6 launches)
    tiny   qkv-from-slabs | o split-K | SiLU-gate-from-slabs | down split-K   (M <= 4: the r02 tiny-batch path, 4 launches)
    rows   qkv (norm on the fly) | o rows+add | SiLU-gate (norm on the fly) | down rows+add                      (4 launches)"""
    dt = torch.bfloat16
    H, I, NQKV = a.hidden, a.inter, a.hidden + 2 * (a.hidden // 4)
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda n, k: [pack_weight_ret((torch.randn(n, k, device="cuda", generator=g) * 0.02).to(dt)) for _ in range(a.copies)]
    wqkv, wo, wug, wdn = mk(NQKV, H), mk(H, H), mk(2 * I, H), mk(H, I)
    nw1 = (1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dt)
    nw2 = (1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dt)
    eps = 1e-5
    for M in [int(v) for v in a.m.split(",")]:
        r0 = torch.randn(M, H, device="cuda", generator=g).to(dt)
        attn = torch.randn(M, H, device="cuda", generator=g).to(dt)       # stands in for the attention output
        act0 = (torch.randn(M, I, device="cuda", generator=g) * 0.5).to(dt)
        row = dict(M=M, mode="layer")
        # ---- equivalence of one pass: old vs rows, starting from the same residual and the same down slabs ----
        def old_pass(res, i, down_part, keep=None):
            pend = add_scale_from_splitk(down_part, res, nw1, eps)
            qkv = linear_splitk(pend.x, wqkv[i], always=True)
            if keep is not None:    # (the slabs live in the shared split-K workspace: the next projection overwrites them)
                keep.append(qkv.slabs[: qkv.k_splits * M * NQKV].clone())
            o_part = linear_splitk(attn, wo[i])
            pend2 = add_scale_from_splitk(o_part, res, nw2, eps)
            act = linear_silu_gate(pend2.x, wug[i], row_scale=pend2)
            return qkv, pend, act, linear_splitk(act, wdn[i])
        def rows_pass(res, i, act_prev):
            linear_rows_add(act_prev, wdn[i], res)
            qkv, pend = linear_splitk_nf(res, nw1, wqkv[i], eps)
            q_slabs = qkv.slabs[: qkv.k_splits * M * NQKV].clone()
            linear_rows_add(attn, wo[i], res)
            act = linear_silu_gate_nf(res, nw2, eps, wug[i])
            return qkv, pend, act, q_slabs
        res_a = r0.clone()
        down_part0 = linear_splitk(act0, wdn[0])
        dp = SplitKPartials(down_part0.slabs[: down_part0.k_splits * M * H].clone(), down_part0.k_splits, M, H, dt)
        kept = []
        qkv_a, pend_a, act_a, _ = old_pass(res_a, 0, dp, kept)
        qa = kept[0]
        torch.cuda.synchronize()
        res_b = r0.clone()
        qkv_b, pend_b, act_b, qb = rows_pass(res_b, 0, act0)
        torch.cuda.synchronize()
        row["ks_qkv"] = qkv_b.k_splits
        row["residual_diff_frac"] = float((res_a != res_b).float().mean())
        row["residual_maxrel"] = float((res_a.float() - res_b.float()).abs().max() / res_a.float().abs().max())
        row["qkv_slabs_bit_equal"] = bool(torch.equal(qa, qb))
        row["qkv_slab_sum_maxrel"] = float((qa.view(qkv_a.k_splits, M, NQKV).sum(0) - qb.view(qkv_b.k_splits, M, NQKV).sum(0)).abs().max()
                                           / qa.view(qkv_a.k_splits, M, NQKV).sum(0).abs().max())
        row["ssq_attn_rel"] = float(((pend_a.ssq.sum(0) - pend_b.ssq.sum(0)).abs() / pend_a.ssq.sum(0)).max())
        row["act_diff_frac"] = float((act_a != act_b).float().mean())
        row["act_maxrel"] = float((act_a.float() - act_b.float()).abs().max() / act_a.float().abs().max())
        # ---- timing ----
        rbuf = r0.clone()
        state = {"down": dp, "act": act0}
        def chain_old(i):
            _, _, _, state["down"] = old_pass(rbuf, i % a.copies, state["down"])
        def chain_rows(i):
            _, _, state["act"], _ = rows_pass_fast(rbuf, i % a.copies, state["act"])
        def rows_pass_fast(res, i, act_prev):
            linear_rows_add(act_prev, wdn[i], res)
            qkv, pend = linear_splitk_nf(res, nw1, wqkv[i], eps)
            linear_rows_add(attn, wo[i], res)
            return qkv, pend, linear_silu_gate_nf(res, nw2, eps, wug[i]), None
        row["old_graph_us"] = round(graph_us(chain_old, a.copies, a.iters), 2)
        row["rows_graph_us"] = round(graph_us(chain_rows, a.copies, a.iters), 2)
        if M <= 4 and tiny_from_splitk_ok(dp, wqkv[0]):
            alt = alt_residual_like(rbuf)
            st2 = {"down": dp}
            def chain_tiny(i):
                j = i % a.copies
                qkv, ssq = linear_splitk_from_splitk(st2["down"], rbuf, alt, nw1, wqkv[j])
                o_part = linear_splitk(attn, wo[j])
                act = linear_silu_gate_from_splitk(o_part, alt, rbuf, nw2, eps, wug[j])
                st2["down"] = linear_splitk(act, wdn[j])
            row["tiny_graph_us"] = round(graph_us(chain_tiny, a.copies, a.iters), 2)
        # single kernels (eager, cycling weights)
        row["down_rows_us"] = round(time_us(lambda i: linear_rows_add(act0, wdn[i % a.copies], rbuf), a.iters), 2)
        row["down_splitk_us"] = round(time_us(lambda i: linear_splitk(act0, wdn[i % a.copies]), a.iters), 2)
        row["o_rows_us"] = round(time_us(lambda i: linear_rows_add(attn, wo[i % a.copies], rbuf), a.iters), 2)
        row["o_splitk_us"] = round(time_us(lambda i: linear_splitk(attn, wo[i % a.copies]), a.iters), 2)
        row["qkv_nf_us"] = round(time_us(lambda i: linear_splitk_nf(rbuf, nw1, wqkv[i % a.copies], eps), a.iters), 2)
        row["qkv_splitk_us"] = round(time_us(lambda i: linear_splitk(rbuf, wqkv[i % a.copies], always=True), a.iters), 2)
        row["silu_nf_us"] = round(time_us(lambda i: linear_silu_gate_nf(rbuf, nw2, eps, wug[i % a.copies]), a.iters), 2)
        row["silu_plain_us"] = round(time_us(lambda i: linear_silu_gate(rbuf, wug[i % a.copies]), a.iters), 2)
        print(json.dumps(row), flush=True)


def pmc_mode(a):
    dt = torch.bfloat16
    N, K = (a.hidden, a.hidden) if a.pmc == "o" else (a.hidden, a.inter)
    M = int(a.m.split(",")[0])
    g = torch.Generator(device="cuda").manual_seed(2)
    ws = [pack_weight_ret((torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dt)) for _ in range(a.copies)]
    x = torch.randn(M, K, device="cuda", generator=g).to(dt)
    res = torch.randn(M, N, device="cuda", generator=g).to(dt)
    us = time_us(lambda i: linear_rows_add(x, ws[i % a.copies], res), a.iters)
    alg = N * K * 2 + M * K * 2 + 2 * M * N * 2          # W once, x once, residual read + write
    print(json.dumps({"kernel": f"gemm_rows_kernel<bf16> ({a.pmc}_proj + residual add)", "M": M, "N": N, "K": K, "us": round(us, 2),
                      "algorithmic_bytes": alg, "TBps": round(alg / us / 1e6, 3)}))


def pack_weight_ret(w):
    pack_weight(w)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", action="store_true", help="the projection side of a whole decode layer, three ways")
    ap.add_argument("--m", default="32,16,8,1")
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=14336)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--no-chain", action="store_true")
    ap.add_argument("--pmc", default="", choices=["", "o", "down"],
                    help="launch only swl_gemm_rows_add at this projection's shape, `--iters` times over `--copies` weights "
                         "(for rocprofv3 --pmc passes: tools/gpu_pmc_rows.sh); prints the algorithmic bytes")
    a = ap.parse_args()
    if a.pmc:
        return pmc_mode(a)
    if a.layer:
        return layer_mode(a)
    dt = torch.bfloat16
    H, I = a.hidden, a.inter
    g = torch.Generator(device="cuda").manual_seed(0)
    wo = [(torch.randn(H, H, device="cuda", generator=g) * 0.02).to(dt) for _ in range(a.copies)]
    for w in wo:
        pack_weight(w)
    wug = [(torch.randn(2 * I, H, device="cuda", generator=g) * 0.02).to(dt) for _ in range(a.copies)]
    wdn = [(torch.randn(H, I, device="cuda", generator=g) * 0.02).to(dt) for _ in range(a.copies)]
    for w in wug + wdn:
        pack_weight(w)
    norm_w = (1.0 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(dt)
    eps = 1e-5
    for M in [int(v) for v in a.m.split(",")]:
        x = torch.randn(M, H, device="cuda", generator=g).to(dt)
        res0 = torch.randn(M, H, device="cuda", generator=g).to(dt)
        row = dict(M=M, N=H, K=H, MB=round(H * H * 2 / 1e6, 1), supported=bool(rows_add_scale_ok(x, wo[0], res0)))
        # ---- correctness: both forms against fp64 --------------------------------------------------------------------
        ref_o = (x.double() @ wo[0].double().t())
        ref_res = (ref_o.to(dt).double() + res0.double())            # (the rounding points of the op, on the exact product)
        r_old = res0.clone()
        part = linear_splitk(x, wo[0])
        assert isinstance(part, SplitKPartials)
        row["ks_old"] = part.k_splits
        p_old = add_scale_from_splitk(part, r_old, norm_w, eps)
        r_new = res0.clone()
        p_new = linear_rows_add_scale(x, wo[0], r_new, norm_w, eps)
        torch.cuda.synchronize()
        scale = float(ref_res.abs().max())
        row["res_err_old"] = float((r_old.double() - ref_res).abs().max()) / scale
        row["res_err_new"] = float((r_new.double() - ref_res).abs().max()) / scale
        row["res_new_vs_old_diff_frac"] = float((r_new != r_old).float().mean())
        row["xs_new_vs_own_residual_bit_equal"] = bool(torch.equal(p_new.x, (r_new.float() * norm_w.float()).to(dt)))
        ssq_ref = (r_new.double() ** 2).sum(1)
        row["ssq_rel_err_new"] = float(((p_new.ssq.double().sum(0) - ssq_ref).abs() / ssq_ref).max())
        # the consumer: SiLU-gate GEMM with 8 / 256 partials of the SAME residual must agree to fp32 rounding of rstd
        ssq8 = (r_new.float() ** 2).view(M, H // 1024, 1024).sum(2).t().contiguous()
        from swiftllm_amd.worker.kernels.rmsnorm import RowScalePending
        a8 = linear_silu_gate(p_new.x, wug[0], row_scale=RowScalePending(p_new.x, ssq8, H // 1024, eps))
        a256 = linear_silu_gate(p_new.x, wug[0], row_scale=p_new)
        torch.cuda.synchronize()
        row["silu_many_vs_8_diff_frac"] = float((a8 != a256).float().mean())
        row["silu_many_vs_8_maxrel"] = float(((a8.float() - a256.float()).abs().max() / a8.float().abs().max()))
        # ---- timing: the op alone ------------------------------------------------------------------------------------
        rbuf = res0.clone()

        def old_pair(i):
            return add_scale_from_splitk(linear_splitk(x, wo[i % a.copies]), rbuf, norm_w, eps)

        def new_one(i):
            return linear_rows_add_scale(x, wo[i % a.copies], rbuf, norm_w, eps)

        row["old_gemm_us"] = round(time_us(lambda i: linear_splitk(x, wo[i % a.copies]), a.iters), 2)
        row["old_pair_us"] = round(time_us(old_pair, a.iters), 2)
        row["new_us"] = round(time_us(new_one, a.iters), 2)
        row["old_pair_graph_us"] = round(graph_us(old_pair, a.copies, a.iters), 2)
        row["new_graph_us"] = round(graph_us(new_one, a.copies, a.iters), 2)
        row["silu8_us"] = round(time_us(lambda i: linear_silu_gate(p_new.x, wug[i % a.copies],
                                                                   row_scale=RowScalePending(p_new.x, ssq8, H // 1024, eps)), a.iters), 2)
        row["silu256_us"] = round(time_us(lambda i: linear_silu_gate(p_new.x, wug[i % a.copies], row_scale=p_new), a.iters), 2)
        # ---- timing: the chain o_proj -> up/gate -> down as the layer runs it, captured ----------------------------------
        if not a.no_chain:
            def chain_old(i):
                pend = add_scale_from_splitk(linear_splitk(x, wo[i % a.copies]), rbuf, norm_w, eps)
                act = linear_silu_gate(pend.x, wug[i % a.copies], row_scale=pend)
                return linear_splitk(act, wdn[i % a.copies])

            def chain_new(i):
                pend = linear_rows_add_scale(x, wo[i % a.copies], rbuf, norm_w, eps)
                act = linear_silu_gate(pend.x, wug[i % a.copies], row_scale=pend)
                return linear_splitk(act, wdn[i % a.copies])

            row["chain_old_graph_us"] = round(graph_us(chain_old, a.copies, a.iters), 2)
            row["chain_new_graph_us"] = round(graph_us(chain_new, a.copies, a.iters), 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
