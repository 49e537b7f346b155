#!/usr/bin/env python3
"""gemm_rows_micro.py — the row-owned decode projections (csrc/gemm_rows.hip: K split inside the workgroup, residual add in the
epilogue) and norm on the fly (csrc/gemm_skinny.hip NF) at Llama-3-8B widths (GPU):
    --layer       the projection side of one decode layer three ways (split-K + consumers | r02 tiny path | rows + NF), each
                  checked against the others first, timed as captured hipGraphs cycling through distinct weight copies;
    --pmc o|down  only swl_gemm_rows_add in a loop (for rocprofv3 --pmc passes, tools/gpu_pmc_rows.sh).
One JSON line per batch size."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker.kernels.linear import (pack_weight, linear_splitk,                                                 linear_silu_gate, SplitKPartials, linear_rows_add, linear_splitk_nf,
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
    raise SystemExit("pick a mode: --layer (projection side of a decode layer, three ways) or --pmc o|down; the r05a o_proj-pair "
                     "mode (swl_gemm_rows_add_scale, removed) is in revision ca4e9a9")


if __name__ == "__main__":
    main()
