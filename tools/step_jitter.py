#!/usr/bin/env python3
"""step_jitter.py — is the decode step time stable from capture to capture and from process to process? (GPU box)

    python tools/step_jitter.py [--batch 32] [--steps 20] [--trials 3] [--no-rows-decode] [--kv-blocks 0]

bench.py's model and KV pool (the product's own sizing at gpu_mem_utilization 0.97 unless --kv-blocks is given), its
decode loop at contexts centred on 1088; per trial the captured graphs are dropped and re-captured. Reports per trial:
wall ms per step (mean / min / max over the steps), GPU ms per step from events around each forward, and what the
caching allocator did during the timed steps (device allocations, retries after an out-of-memory, reserved bytes, free
HBM) — a step that pays hipMalloc/hipFree because the pool sizing left no headroom shows up there, not in a kernel trace."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--kv-blocks", type=int, default=0)
    ap.add_argument("--no-rows-decode", action="store_true")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--batch", str(a.batch), "--kv-blocks", str(a.kv_blocks)] + (["--no-rows-decode"] if a.no_rows_decode else [])
    args = bench.parse_args()
    cfg = bench.model_config_dict("llama3-8b")
    model = bench.build_model(args, cfg, a.batch * 80 + 8, a.batch, 1300, True)
    free0, total = torch.cuda.mem_get_info()
    print(json.dumps(dict(event="built", kv_blocks=int(model.num_blocks), free_gb=round(free0 / 1e9, 3),
                          reserved_gb=round(torch.cuda.memory_reserved() / 1e9, 3),
                          allocated_gb=round(torch.cuda.memory_allocated() / 1e9, 3))), flush=True)
    for trial in range(a.trials):
        model._drop_decode_graphs()
        model._lookahead = None
        run = bench.DecodeRun(model, a.batch, cfg["vocab_size"], seed=3 + trial)
        run.jump_to(1088 - a.steps // 2 - 8)
        for _ in range(8):
            run.step()
        torch.cuda.synchronize()
        st0 = torch.cuda.memory_stats()
        walls, gpus = [], []
        for _ in range(a.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            run.step()
            e1.record()
            e1.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
            gpus.append(e0.elapsed_time(e1))
        st1 = torch.cuda.memory_stats()
        free1, _ = torch.cuda.mem_get_info()
        run.release()
        walls_s = sorted(walls)
        print(json.dumps(dict(trial=trial, rows_decode=not a.no_rows_decode, batch=a.batch,
                              wall_ms_mean=round(sum(walls) / len(walls), 4), wall_ms_min=round(walls_s[0], 4),
                              wall_ms_median=round(walls_s[len(walls_s) // 2], 4), wall_ms_max=round(walls_s[-1], 4),
                              gpu_ms_mean=round(sum(gpus) / len(gpus), 4), gpu_ms_min=round(min(gpus), 4), gpu_ms_max=round(max(gpus), 4),
                              device_allocs=st1["num_device_alloc"] - st0["num_device_alloc"],
                              device_frees=st1["num_device_free"] - st0["num_device_free"],
                              alloc_retries=st1["num_alloc_retries"] - st0["num_alloc_retries"],
                              ooms=st1["num_ooms"] - st0["num_ooms"], free_gb=round(free1 / 1e9, 3),
                              reserved_gb=round(torch.cuda.memory_reserved() / 1e9, 3),
                              graphs=len(model._decode_graphs))), flush=True)


if __name__ == "__main__":
    main()
