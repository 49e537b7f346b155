#!/usr/bin/env python3
"""gemm_wgk_micro.py — in-workgroup split-K projections (csrc/gemm_wgk.hip) against the cross-workgroup split-K path
they replace, on the GPU, per shape and batch:

  o / down : [swl_gemm_skinny_packed_partial ; swl_splitk_add_scale]  vs  swl_gemm_wgk_add_scale
  qkv      : swl_gemm_skinny_packed_partial (slabs for the attention prologue)  vs  swl_gemm_wgk (one fp32 slab)

Each variant is captured into a hipGraph of `--chain` launches that cycle through distinct weight copies (far past the
256 MiB Infinity Cache) and replayed: GPU time per launch including the kernel boundaries a decode step pays, no host
launch cost. The first launch of every variant is checked: residual / x_scaled bit-equal between the two paths (same
K partition, same order), row sums of squares equal to fp32 rounding, qkv against an fp64 matmul."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd import _hip

SHAPES = {"o": (4096, 4096), "qkv": (6144, 4096), "down": (4096, 14336)}


def graph_time(fn, chain, reps):
    """us per call of fn(i), i = 0..chain-1 captured once, replayed `reps` times"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(chain):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
    torch.cuda.current_stream().wait_stream(s)
    return e0.elapsed_time(e1) * 1e3 / (reps * chain)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--ms", default="32,8,1")
    ap.add_argument("--shapes", default="o,qkv,down")
    ap.add_argument("--chain", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dtype = getattr(torch, a.dtype)
    code = _hip.dtype_code(dtype)
    lib = _hip.load()
    for name in a.shapes.split(","):
        N, K = SHAPES[name]
        copies = max(4, min(16, int(2.4e9 // (N * K * 2))))
        g = torch.Generator(device="cuda").manual_seed(5)
        wps, w0 = [], None
        for c in range(copies):
            w = torch.empty(N, K, dtype=dtype, device="cuda").normal_(0, 0.02, generator=g)
            wp = torch.empty_like(w)
            _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, _hip.stream())
            wps.append(wp)
            if c == 0:
                w0 = w
        ks = lib.swl_gemm_skinny_choose_splits(N, K)
        for M in [int(v) for v in a.ms.split(",")]:
            x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
            res0 = torch.randn(M, N, device="cuda", generator=g).to(dtype)
            nw = (1 + 0.1 * torch.randn(N, device="cuda", generator=g)).to(dtype)
            slabs = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
            out = {"shape": name, "M": M, "N": N, "K": K, "MB": round(N * K * 2 / 1e6, 1), "ks_old": ks,
                   "supported": bool(lib.swl_gemm_wgk_supported(M, N, K))}
            if name == "qkv":
                q32 = torch.empty(M, N, dtype=torch.float32, device="cuda")

                def old(i):
                    _hip.call("swl_gemm_skinny_packed_partial", slabs.data_ptr(), slabs.numel() * 4, x.data_ptr(),
                              wps[i % copies].data_ptr(), M, N, K, K, ks, code, _hip.stream())

                def new(i):
                    _hip.call("swl_gemm_wgk", q32.data_ptr(), 1, x.data_ptr(), wps[i % copies].data_ptr(), None, 0, 0,
                              0.0, M, N, K, K, N, code, _hip.stream())
                old(0); new(0)
                torch.cuda.synchronize()
                ref = x.double() @ w0.double().t()
                got_old = slabs[: ks * M * N].view(ks, M, N).sum(0).double()
                out["err_old"] = float((got_old - ref).abs().max())
                out["err_new"] = float((q32.double() - ref).abs().max())
                out["ref_absmax"] = float(ref.abs().max())
            else:
                parts_old = N // 1024
                r_old, r_new = res0.clone(), res0.clone()
                xs_old, xs_new = torch.empty_like(res0), torch.empty_like(res0)
                ssq_old = torch.zeros(parts_old, M, dtype=torch.float32, device="cuda")
                ssq_new = torch.zeros(N // 32, 32, dtype=torch.float32, device="cuda")

                def old(i):
                    _hip.call("swl_gemm_skinny_packed_partial", slabs.data_ptr(), slabs.numel() * 4, x.data_ptr(),
                              wps[i % copies].data_ptr(), M, N, K, K, ks, code, _hip.stream())
                    _hip.call("swl_splitk_add_scale", xs_old.data_ptr(), r_old.data_ptr(), nw.data_ptr(), slabs.data_ptr(),
                              ks, ssq_old.data_ptr(), M, N, code, _hip.stream())

                def new(i):
                    _hip.call("swl_gemm_wgk_add_scale", xs_new.data_ptr(), r_new.data_ptr(), ssq_new.data_ptr(),
                              nw.data_ptr(), x.data_ptr(), wps[i % copies].data_ptr(), M, N, K, K, code, _hip.stream())
                old(0); new(0)
                torch.cuda.synchronize()
                out["residual_bit_equal"] = bool(torch.equal(r_old, r_new)) if ks == 8 else None
                out["xs_bit_equal"] = bool(torch.equal(xs_old, xs_new)) if ks == 8 else None
                out["residual_maxdiff"] = float((r_old.float() - r_new.float()).abs().max())
                so, sn = ssq_old.sum(0), ssq_new[:, :M].sum(0)
                out["ssq_rel_err"] = float(((so - sn).abs() / so).max())
                ref = (x.double() @ w0.double().t() + res0.double())
                out["err_new_vs_f64"] = float((r_new.double() - ref).abs().max())
                r_old.copy_(res0); r_new.copy_(res0)
            out["old_us"] = round(graph_time(old, a.chain, a.reps), 2)
            out["new_us"] = round(graph_time(new, a.chain, a.reps), 2) if out["supported"] else None
            out["old_TBps"] = round(N * K * 2 / out["old_us"] / 1e6, 2)
            if out["new_us"]:
                out["new_TBps"] = round(N * K * 2 / out["new_us"] / 1e6, 2)
            print(json.dumps(out), flush=True)
        del wps


if __name__ == "__main__":
    main()
