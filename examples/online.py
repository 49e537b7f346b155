"""Online serving through the Engine (tokenize -> schedule -> forward loop), the flow of the reference's
examples/online.py: several requests arrive at once, one of them is streamed token by token.

    python examples/online.py --model-path /path/to/llama [--dtype bfloat16] [--piggyback]

Needs a HuggingFace tokenizer in the model directory unless --token-ids is given (then random prompts of
token ids are served, e.g. against a random-init checkpoint from oracle/synth.py).
"""
import argparse
import asyncio
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swiftllm  # noqa: E402

PROMPTS = ["Life blooms like a flower, far away", "one two three four five", "To be or not to be,"]


async def main(args):
    cfg = swiftllm.EngineConfig(model_path=args.model_path, use_dummy=False, block_size=16,
                                gpu_mem_utilization=0.9, num_cpu_blocks=1024, max_seqs_in_block_table=256,
                                max_blocks_per_seq=2048, max_batch_size=64, max_tokens_in_batch=16384,
                                dtype=args.dtype, use_hip_graph=True)
    engine = swiftllm.Engine(cfg, piggyback=args.piggyback)
    await engine.initialize()
    loops = asyncio.ensure_future(engine.start_all_event_loops())

    if args.token_ids:
        rng = random.Random(0)
        vocab = engine.model_config.vocab_size
        raws = [swiftllm.RawRequest("", args.output_len, [rng.randrange(vocab) for _ in range(n)]) for n in (9, 5, 22)]
    else:
        raws = [swiftllm.RawRequest(p, args.output_len) for p in PROMPTS]

    async def streamed(raw):
        async for step in engine.add_request_and_stream(raw):
            piece = step.token_id if args.token_ids else await engine.tokenization_engine.decode([step.token_id])
            print(f"  [stream] {piece!r}", flush=True)

    t0 = time.perf_counter()
    results = await asyncio.gather(streamed(raws[0]), *(engine.add_request_and_wait(r) for r in raws[1:]))
    dt = time.perf_counter() - t0
    for _, token_ids in results[1:]:
        print(token_ids if args.token_ids else await engine.tokenization_engine.decode(token_ids))
    total = args.output_len * len(raws)
    print(f"{total} tokens in {dt:.2f} s ({engine.num_forwards} forwards)")
    loops.cancel()


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--output-len", type=int, default=32)
    ap.add_argument("--piggyback", action="store_true")
    ap.add_argument("--token-ids", action="store_true")
    asyncio.run(main(ap.parse_args()))
