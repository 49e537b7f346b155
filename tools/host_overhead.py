#!/usr/bin/env python3
"""host_overhead.py — where does a decode step's wall time go on the HOST side? (GPU box)

    SWL_HOST_PROFILE=1 python tools/host_overhead.py [--batch 32] [--steps 64]

Runs bench.py's decode loop (Llama-3-8B dims, bf16, contexts centred at 1088) and prints the mean host time of each
section of LlamaModel.forward (plan / blocks / upload / launch / wait for the tokens) next to the step time, with hipGraph
replay and with eager launches. "wait_tokens" is GPU time the host sleeps through; everything else is time the GPU may idle.
"""
import argparse, json, os, sys
os.environ.setdefault("SWL_HOST_PROFILE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=64)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--batch", str(a.batch), "--kv-blocks", "4096"]
    args = bench.parse_args()
    cfg = bench.model_config_dict("llama3-8b")
    model = bench.build_model(args, cfg, a.batch * 80 + 8, a.batch, 1300, True)
    for graph in (True, False):
        model.engine_config.use_hip_graph = graph
        run = bench.DecodeRun(model, a.batch, cfg["vocab_size"], seed=3)
        run.jump_to(1088 - a.steps // 2 - 8)
        for _ in range(8):
            run.step()
        model.host_profile()
        dt, first, last = run.timed_steps(0, a.steps)
        prof = model.host_profile()
        run.release()
        print(json.dumps(dict(hip_graph=graph, batch=a.batch, ms_per_step=round(dt / a.steps * 1e3, 4), contexts=[first, last],
                              host_us={k: round(v * 1e6, 1) for k, v in prof.items()})))


if __name__ == "__main__":
    main()
