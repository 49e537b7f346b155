#!/usr/bin/env python3
"""rows_kernel_us.py — swl_gemm_rows_add alone (o_proj / down_proj shapes of Llama-3-8B), timed as a captured hipGraph of
`--copies` launches over distinct weight copies (no host launch rate in the number). One JSON line per (projection, M).
SWL_ROWS_X=frag selects the r05 fragment-shaped x loads (A/B)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker.kernels.linear import pack_weight, linear_rows_add, linear_splitk
from tools.gemm_rows_micro import graph_us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", default="32,16,8,1")
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=14336)
    ap.add_argument("--copies", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(3)
    for name, (N, K) in (("o", (a.hidden, a.hidden)), ("down", (a.hidden, a.inter))):
        ws = []
        for _ in range(a.copies):
            w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(dt)
            pack_weight(w)
            ws.append(w)
        for M in [int(v) for v in a.m.split(",")]:
            x = torch.randn(M, K, device="cuda", generator=g).to(dt)
            res = torch.randn(M, N, device="cuda", generator=g).to(dt)
            rows = graph_us(lambda i: linear_rows_add(x, ws[i], res), a.copies, a.iters)
            sk = graph_us(lambda i: linear_splitk(x, ws[i]), a.copies, a.iters)
            print(json.dumps({"proj": name, "M": M, "N": N, "K": K, "rows_us": round(rows, 2), "splitk_gemm_only_us": round(sk, 2),
                              "x_form": os.environ.get("SWL_ROWS_X", "lds")}), flush=True)


if __name__ == "__main__":
    main()
