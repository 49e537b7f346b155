#!/usr/bin/env python3
"""engine_overhead.py — host cost of one Engine iteration (scheduler + forward hand-off + token fan-out), measured on
CPU with a stand-in data plane: `forward` "launches" (fires after_launch_hook), then sleeps --step-ms with the GIL
released — what LlamaModel.forward does while it waits for the sampled tokens — and returns constant tokens.

    python tools/engine_overhead.py [--requests 64] [--gen-len 128] [--max-batch 32] [--step-ms 4.0] [--no-hook]

Prints us per iteration above the simulated step time (R requests of 1024 prompt tokens arrive at once, streamed;
includes request creation and admission) and the gap between one decode forward's return and the next one's entry —
the time the GPU would idle per step."""
import argparse, asyncio, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd import Engine, EngineConfig, RawRequest  # noqa: E402


class StandIn:
    num_blocks = 120000

    def __init__(self, step_s, hook):
        import types
        self.model_config = types.SimpleNamespace(vocab_size=128256)
        self.step_s = step_s
        if hook:
            self.after_launch_hook = None
        self.busy = 0.0
        self.last_end = None
        self.gaps = []      # seconds between one forward's return and the next one's entry (pure-decode steps)

    def forward(self, ids, seq_ids, lens):
        if self.last_end is not None and len(lens) == len(ids):
            self.gaps.append(time.perf_counter() - self.last_end)
        hook = getattr(self, "after_launch_hook", None)
        if hook is not None:
            hook()
        t = time.perf_counter()
        if self.step_s:
            time.sleep(self.step_s)
        self.busy += time.perf_counter() - t
        self.last_end = time.perf_counter()
        return [7] * len(ids)

    def swap_in_seqs(self, s): pass
    def swap_out_seqs(self, s): pass
    def free_seqs_resources(self, s): pass


async def run(a, hook):
    ec = EngineConfig(model_path="", use_dummy=True, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=1024,
                      max_seqs_in_block_table=1024, max_blocks_per_seq=128, max_batch_size=a.max_batch,
                      max_tokens_in_batch=a.max_batch * 1024)
    m = StandIn(a.step_ms * 1e-3, hook)
    eng = Engine(ec, model=m, piggyback=True)
    await eng.initialize()
    loops = asyncio.ensure_future(eng.start_all_event_loops())

    async def one():
        async for _ in eng.add_request_and_stream(RawRequest("", a.gen_len, [1] * 1024)):
            pass
    t0 = time.perf_counter()
    await asyncio.gather(*(one() for _ in range(a.requests)))
    dt = time.perf_counter() - t0
    loops.cancel()
    return {"after_launch_hook": hook, "requests": a.requests, "gen_len": a.gen_len, "max_batch": a.max_batch,
            "step_ms": a.step_ms, "forwards": eng.num_forwards, "wall_s": round(dt, 4),
            "host_us_per_iteration": round((dt - m.busy) / eng.num_forwards * 1e6, 1),
            "decode_gap_us_p50": round(sorted(m.gaps)[len(m.gaps) // 2] * 1e6, 1),
            "decode_gap_us_mean": round(sum(m.gaps) / len(m.gaps) * 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=64)
    ap.add_argument("--gen-len", type=int, default=128)
    ap.add_argument("--max-batch", type=int, default=32)
    ap.add_argument("--step-ms", type=float, default=4.0)
    ap.add_argument("--no-hook", action="store_true")
    a = ap.parse_args()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        res = asyncio.run(run(a, not a.no_hook))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
