#!/usr/bin/env python3
"""blas_block_planner.py — would a measured cost table + shortest-path split of the token count beat the fixed
row-block rule of swiftllm_amd/worker/kernels/linear.py (_blas_linear)?

1. calibrate: time one layer's four projections (torch.nn.functional.linear -> hipBLASLt) at every multiple of --step
   rows up to --max-rows (HIP events, --iters launches each);
2. plan: best[M] = min over table sizes b <= M of cost[b] + best[M - b] (a remainder below --step is priced as one
   --step-row call: an upper bound);
3. check: for --check token counts, time (a) one call, (b) the shipped fixed rule, (c) the planned split, for real.

    python tools/blas_block_planner.py [--model llama3-8b] [--check 2816,3328,4124,5500,6500,7500,10240,13000]

`plan()` is pure arithmetic (tests/test_host_logic.py runs it on a synthetic table)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {
    "llama3-8b": {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336)},
    "llama2-7b": {"qkv": (12288, 4096), "o": (4096, 4096), "up_gate": (22016, 4096), "down": (4096, 11008)},
}


def plan(cost: dict, m: int, step: int):
    """cost: {rows: us} on a grid of multiples of `step`. Returns (predicted us, [block sizes]) for m rows: the cheapest
    way to cover m with table sizes, the last (partial) block priced as the next grid point up."""
    sizes = sorted(cost)
    units = -(-m // step)                       # m rounded up to the grid
    best = [0.0] + [float("inf")] * units
    take = [0] * (units + 1)
    for u in range(1, units + 1):
        for b in sizes:
            bu = b // step
            if bu > u:
                break
            c = best[u - bu] + cost[b]
            if c < best[u]:
                best[u], take[u] = c, b
    blocks, u = [], units
    while u > 0:
        blocks.append(take[u])
        u -= take[u] // step
    blocks.sort(reverse=True)
    over = sum(blocks) - m                      # the grid overshoot comes off the smallest block
    if over:
        blocks[-1] -= over
    return best[units], blocks


def fixed_rule(m: int):
    """The split _blas_linear ships (kept in step with kernels/linear.py by tests/test_host_logic.py)."""
    if m <= 4096 or (m >= 16384 and m % 4096 == 0):
        return [m]
    blocks, left = [], m
    if m >= 16384:
        blocks.append(m // 4096 * 4096)
        left -= blocks[0]
    for size in (8192, 4096):
        while left >= size:
            blocks.append(size)
            left -= size
    if left:
        blocks.append(left)
    return blocks


def main():
    import torch
    import torch.nn.functional as F
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b", choices=sorted(SHAPES))
    ap.add_argument("--step", type=int, default=256)
    ap.add_argument("--max-rows", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--check", default="2816,3328,4124,5500,6500,7500,10240,13000")
    a = ap.parse_args()
    dev = "cuda"
    w = {k: torch.randn(n, kk, device=dev, dtype=torch.bfloat16) * 0.02 for k, (n, kk) in SHAPES[a.model].items()}
    xs = {kk: torch.randn(max(a.max_rows, 16384), kk, device=dev, dtype=torch.bfloat16)
          for kk in {kk for _, kk in SHAPES[a.model].values()}}

    def layer_us(blocks):
        """one layer's four projections over sum(blocks) rows, each projection taken block by block"""
        m = sum(blocks)
        outs = {k: torch.empty(m, n, device=dev, dtype=torch.bfloat16) for k, (n, _) in SHAPES[a.model].items()}

        def run():
            for k, (n, kk) in SHAPES[a.model].items():
                s = 0
                for b in blocks:
                    torch.mm(xs[kk][s:s + b], w[k].t(), out=outs[k][s:s + b])
                    s += b
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    cost = {m: layer_us([m]) for m in range(a.step, a.max_rows + 1, a.step)}
    print(json.dumps({"model": a.model, "cost_table_us_per_layer": {str(k): round(v, 1) for k, v in cost.items()}}), flush=True)
    for m in [int(x) for x in a.check.split(",")]:
        pred, blocks = plan(cost, m, a.step)
        row = {"M": m, "one_call_us": round(layer_us([m]), 1), "fixed_rule": fixed_rule(m),
               "fixed_rule_us": round(layer_us(fixed_rule(m)), 1), "planned": blocks, "planned_predicted_us": round(pred, 1),
               "planned_us": round(layer_us(blocks), 1)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
