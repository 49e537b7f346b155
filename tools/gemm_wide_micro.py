#!/usr/bin/env python3
"""gemm_wide_micro.py — swl_gemm_packed_wide / _silu_gate (csrc/gemm_wide.hip) against hipBLASLt at the four projection
shapes of a Llama-3-8B decode layer, for decode batches of 65..256 tokens (GPU). Checks every variant against an fp32
reference first, then times it with HIP events over launches that cycle through 8 distinct weight copies (far past the
256 MiB Infinity Cache for the big shapes). One JSON line per (shape, M)."""
import argparse, hashlib, json, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd import _hip
from swiftllm_amd.worker.kernels.linear import pack_weight, _workspace

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336), "lm_head": (128256, 4096)}


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


def sha(t):
    """Digest of a result tensor's bits (runs with different SWL_WIDE_TS / variant libraries must agree on it)."""
    return hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="96,128,192,256")
    ap.add_argument("--shapes", default="qkv,o,up_gate,down")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--cycle-mb", type=float, default=1200.0, help="minimum total size of the weight copies cycled through")
    ap.add_argument("--auto-only", action="store_true", help="the library's own plan only, no hipBLASLt timing (for PMC passes)")
    a = ap.parse_args()
    dtype = getattr(torch, a.dtype)
    code = _hip.dtype_code(dtype)
    g = torch.Generator(device="cuda").manual_seed(0)
    tag = dict(ts=os.environ.get("SWL_WIDE_TS", "auto"), lib=os.path.basename(os.environ.get("SWIFTLLM_HIP_LIB", "product")))
    for name in a.shapes.split(","):
        N, K = SHAPES[name]
        # cycle through enough copies that a re-read never finds its lines in the 256 MiB Infinity Cache (r06d: with 6 copies
        # of the 34 MB o_proj weight, plain loads looked 10 % faster than non-temporal ones — an artefact of 204 MB of weights)
        copies = max(a.copies, -(-int(a.cycle_mb * 1e6) // (N * K * 2))) if N * K * 2 < (1 << 30) else 2
        ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dtype) for _ in range(copies)]
        wps = [pack_weight(w) for w in ws]
        for M in [int(x) for x in a.m.split(",")]:
            x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
            ref = x.float() @ ws[0].float().t()
            row = dict(shape=name, M=M, N=N, K=K, MB=round(N * K * 2 / 1e6, 1), copies=copies, **tag)
            if not a.auto_only:
                row["blas_us"] = round(time_us(lambda i: F.linear(x, ws[i % copies]), a.iters), 2)
            out = torch.empty(M, N, device="cuda", dtype=dtype)
            scratch = _workspace(x.device, 16 * M * N * 4)
            variants = [(0, 0)] + ([] if a.auto_only else [(nw, ks) for nw in (4, 8) for ks in (1, 2, 4, 8, 16)])
            for nw, ks in variants:
                if ks and (K % (64 * ks) or K // ks < 256):
                    continue
                def run(i, nw=nw, ks=ks):
                    _hip.call("swl_gemm_packed_wide", out.data_ptr(), x.data_ptr(), wps[i % copies].data_ptr(),
                              scratch.data_ptr(), scratch.numel() * 4, M, N, K, K, N, nw, ks, code, _hip.stream())
                out.zero_()
                run(0)
                err = float((out.float() - ref).abs().max() / ref.abs().max())
                key = f"w{nw}k{ks}" if nw else "auto"
                row[key + "_us"] = round(time_us(run, a.iters), 2)
                row[key + "_relerr"] = round(err, 5)
                run(0)
                row[key + "_sha"] = sha(out)
            if name == "up_gate":
                I = N // 2
                og = torch.empty(M, I, device="cuda", dtype=dtype)
                r2 = F.linear(x, ws[0])
                want = (r2[:, :I].float() * F.silu(r2[:, I:].float())).to(dtype)

                def blas_pair(i):
                    r = F.linear(x, ws[i % copies])
                    return r[:, :I] * F.silu(r[:, I:])
                if not a.auto_only:
                    row["blas_plus_silu_us"] = round(time_us(blas_pair, a.iters), 2)
                for nw in ((0,) if a.auto_only else (4, 8)):
                    def run(i, nw=nw):
                        _hip.call("swl_gemm_packed_wide_silu_gate", og.data_ptr(), x.data_ptr(), wps[i % copies].data_ptr(),
                                  M, I, K, K, I, nw, code, _hip.stream())
                    og.zero_()
                    run(0)
                    row[f"silu_w{nw}_relerr"] = round(float((og.float() - want.float()).abs().max() / want.float().abs().max()), 5)
                    row[f"silu_w{nw}_us"] = round(time_us(run, a.iters), 2)
                    run(0)
                    row[f"silu_w{nw}_sha"] = sha(og)
            print(json.dumps(row), flush=True)
        del ws, wps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
