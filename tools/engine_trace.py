"""Phase anatomy of the persistent one-sequence decode step (csrc/decode_engine.hip): CU 0's consumer wave 0 stamps the
100 MHz wall clock at eight points per layer (debug_stamps of swl_decode_engine_step); this prints the mean duration of
every phase over the layers and steps, next to the whole step's wall time.

    python tools/engine_trace.py [--layers 32] [--context 1088] [--steps 20] [--dtype bfloat16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

STAMPS = ["layer start", "P0 r gathered", "P0 normed (+KV requested)", "P1 qkv gemv done", "P1 qkv published",
          "P2 q/k/v gathered", "P2 attended", "P2 partials published", "P2b partials merged-in", "P3 o_attn gathered",
          "P3 o gemv done + r published", "P4 r gathered", "P4 normed", "P4 up/gate gemv done + act published",
          "P5 act gathered", "P5 down gemv done + r published"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--context", type=int, default=1088)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bfloat16")
    a = ap.parse_args()
    import torch
    cfg = bench.model_config_dict("llama3-8b")
    cfg["num_hidden_layers"] = a.layers
    args = argparse.Namespace(dtype=a.dtype, fuse_qkv=True, skinny_gemm=True, splitk_fusion=True, decode_engine=True, kv_blocks=4096,
                              kv_placement="bottom")
    model = bench.build_model(args, cfg, 256, 1, a.context + a.steps + 64, True)
    assert model._engine is not None, "engine not available for this shape / device"
    run = bench.DecodeRun(model, 1, cfg["vocab_size"], 7)
    run.jump_to(a.context)
    for _ in range(6):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    # stamps: eager launches (a captured graph has the NULL pointer baked in)
    model.engine_config.use_hip_graph = False
    model._drop_decode_graphs()
    model._lookahead = None
    stamps = model._engine.enable_debug_stamps()
    seg = torch.zeros(16, dtype=torch.float64)      # mean time from the previous stamp to stamp k (k = 0: from the previous layer's end)
    spread = torch.zeros(16, dtype=torch.float64)   # mean (max - min) over the traced CUs of the time they reach stamp k
    layer_total = 0.0
    n = 0
    for _ in range(a.steps):
        run.step()
        torch.cuda.synchronize()
        s = stamps.cpu().to(torch.float64) / 100.0      # [7 CUs][L][16] us
        d = s[:, :, 1:] - s[:, :, :-1]
        seg[1:] += d.mean(dim=(0, 1))
        if a.layers > 1:
            seg[0] += (s[:, 1:, 0] - s[:, :-1, 15]).mean()
            layer_total += float((s[:, 1:, 0] - s[:, :-1, 0]).mean())
        else:
            layer_total += float((s[:, 0, 15] - s[:, 0, 0]).mean())
        spread += (s.max(dim=0).values - s.min(dim=0).values).mean(dim=0)
        n += 1
    out = {"ms_per_step_graph": round(wall, 4), "layers": a.layers, "context": a.context, "dtype": a.dtype,
           "engine_flags": model._engine.flags, "us_per_layer": round(layer_total / n, 2),
           "segment_us": {name: round(float(v) / n, 2) for name, v in zip(STAMPS, seg)},
           "arrival_spread_over_7_cus_us": {name: round(float(v) / n, 2) for name, v in zip(STAMPS, spread)}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
