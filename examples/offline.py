"""Offline generation straight on the data plane (no engine, no scheduler) — the flow of the reference's
examples/offline.py against this implementation: build the model, size the KV pool from free HBM,
prefill a few prompts in one batch, then decode greedily step by step.

    python examples/offline.py --model-path /path/to/llama [--dtype bfloat16] [--steps 20]

With a HuggingFace tokenizer in the model directory the prompts are text; otherwise (e.g. the random-init
checkpoints written by oracle/synth.py) random token ids are used and ids are printed.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swiftllm  # noqa: E402

PROMPTS = ["Life blooms like a flower, far away", "one two three four five",
           "A B C D E F G H I J K L M N O P Q R S T U V", "To be or not to be,"]


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--use-dummy", action="store_true")
    args = ap.parse_args()

    cfg = swiftllm.EngineConfig(model_path=args.model_path, use_dummy=args.use_dummy, block_size=16,
                                gpu_mem_utilization=0.9, num_cpu_blocks=0, max_seqs_in_block_table=128,
                                max_blocks_per_seq=2048, max_batch_size=16, max_tokens_in_batch=2048 * 16,
                                dtype=args.dtype, use_hip_graph=True)
    t0 = time.perf_counter()
    model = swiftllm.LlamaModel(cfg)
    model.load_weights()
    num_blocks = model.profile_num_blocks()
    model.init_kvcache_and_swap(min(num_blocks, 4096))
    print(f"model ready in {time.perf_counter() - t0:.2f} s; {num_blocks} KV blocks fit, using {model.num_blocks}")

    tokenizer = None
    if any(n.startswith("tokenizer") for n in os.listdir(args.model_path)):
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(args.model_path)
        input_ids = tokenizer(PROMPTS)["input_ids"]
    else:
        import random
        rng = random.Random(0)
        vocab = model.model_config.vocab_size
        input_ids = [[rng.randrange(vocab) for _ in range(n)] for n in (9, 5, 22, 6)]

    seq_ids = list(range(len(input_ids)))
    outputs = [model.forward(input_ids, seq_ids, [])]
    lens = [len(x) for x in input_ids]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lens = [n + 1 for n in lens]
        outputs.append(model.forward([[t] for t in outputs[-1]], seq_ids, lens))
    dt = time.perf_counter() - t0
    print(f"{args.steps} decode steps x {len(seq_ids)} sequences: {dt / args.steps * 1e3:.2f} ms/step")
    for i in seq_ids:
        toks = [step[i] for step in outputs]
        print(f"[{i}]", tokenizer.decode(toks, skip_special_tokens=True) if tokenizer else toks)
    model.free_seqs_resources(seq_ids)


if __name__ == "__main__":
    main()
