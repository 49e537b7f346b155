"""Drop-in acceptance (north_star: "examples/offline.py and the API server run unchanged"):
the REFERENCE's own example scripts — staged byte for byte by `python -m oracle.make_ref` into oracle/_ref/examples/,
their sha256 checked against the staging manifest — are executed as they are, in a subprocess, with `import swiftllm`
resolving to this repository's alias package, on a synthetic checkpoint (random-init 2-layer LLaMA + a locally built
word-level tokenizer that `transformers.AutoTokenizer` loads without a network). What they print must be what the CPU
oracle generates greedily for the same prompts.

  * examples/offline.py: LlamaModel driven directly — EngineConfig keyword construction (no dtype: fp16), load_weights,
    profile_num_blocks at gpu_mem_utilization 0.99 (the whole 288 GB: tens of millions of 8 KiB blocks for this small
    model), init_kvcache_and_swap, one prefill forward of 4 prompts and 20 decode forwards;
  * examples/online.py: Engine (tokenizer, scheduler, swap pool of 1024 blocks), streaming and non-streaming requests.
"""
import hashlib
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from oracle import synth
from oracle.ref_model import RefLlamaModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "oracle", "_ref", "examples")
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900),
              pytest.mark.skipif(not os.path.isfile(os.path.join(EX, "offline.py")),
                                 reason="oracle/_ref/examples not staged (python -m oracle.make_ref)")]


def _staged_script(name):
    path = os.path.join(EX, name)
    with open(os.path.join(ROOT, "oracle", "_ref", "MANIFEST.json"), encoding="utf-8") as f:
        want = json.load(f)["examples"]["examples/" + name]
    with open(path, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == want, f"{path} is not the reference's file"
    return path


@pytest.fixture(scope="module")
def checkpoint(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("example_model"))
    # offline.py forges a 16 x 2048-token prefill when it profiles: the rotary table must cover it
    cfg = synth.make_config(max_position_embeddings=4096)
    sd = synth.make_state_dict(cfg, seed=3)
    synth.write_model_dir(path, cfg, sd)
    synth.write_tokenizer(path, cfg["vocab_size"])
    return path, cfg, sd


def _oracle_generation(cfg, sd, prompt_ids, new_tokens):
    """Greedy continuation of every prompt on its own (greedy decoding does not depend on batching)."""
    from swiftllm_amd import EngineConfig, LlamaModelConfig
    ec = EngineConfig(model_path="", use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
                      max_seqs_in_block_table=4, max_blocks_per_seq=16, max_batch_size=1, max_tokens_in_batch=256)
    out = []
    for ids, n in zip(prompt_ids, new_tokens):
        ref = RefLlamaModel(LlamaModelConfig(cfg), ec, sd, torch.float16)
        ref.init_kvcache_and_swap(16)
        toks = ref.forward([ids], [0], [])
        cur = len(ids)
        while len(toks) < n:
            cur += 1
            toks += ref.forward([[toks[-1]]], [0], [cur])
        out.append(toks[:n])
    return out


def _run(script, args):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1",
               TOKENIZERS_PARALLELISM="false", HF_HUB_OFFLINE="1", TRANSFORMERS_OFFLINE="1")
    r = subprocess.run([sys.executable, script] + args, cwd=os.path.dirname(script), env=env, capture_output=True,
                       text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + "\n---- stderr ----\n" + r.stderr[-5000:]
    return r.stdout


def test_reference_offline_example_runs_unchanged(checkpoint):
    from transformers import AutoTokenizer
    path, cfg, sd = checkpoint
    out = _run(_staged_script("offline.py"), ["--model-path", path])
    tok = AutoTokenizer.from_pretrained(path)
    prompt_ids = tok(synth.EXAMPLE_PROMPTS)["input_ids"]
    want = _oracle_generation(cfg, sd, prompt_ids, [21] * 4)       # the prompt pass + 20 decode rounds
    m = re.search(r"Number of blocks: (\d+)", out)
    assert m and int(m.group(1)) > 1_000_000, out[:400]            # the whole GPU went to the pool
    for prompt, toks in zip(synth.EXAMPLE_PROMPTS, want):
        line = f"{prompt}|{tok.decode(toks, skip_special_tokens=True)}"
        assert line in out.splitlines(), (line, out[-1500:])


@pytest.mark.parametrize("streaming", [False, True], ids=["non_streaming", "streaming"])
def test_reference_online_example_runs_unchanged(checkpoint, streaming):
    from transformers import AutoTokenizer
    path, cfg, sd = checkpoint
    out = _run(_staged_script("online.py"), ["--model-path", path] + (["--streaming"] if streaming else []))
    tok = AutoTokenizer.from_pretrained(path)
    lens = [10, 50, 5, 15]                                          # online.py:66-71
    prompt_ids = tok(synth.EXAMPLE_PROMPTS)["input_ids"]
    want = _oracle_generation(cfg, sd, prompt_ids, lens)
    blocks = out.split("---------------------------------")
    for prompt, toks in zip(synth.EXAMPLE_PROMPTS, want):
        mine = [b for b in blocks if f"Prompt: {prompt}\n" in b]
        assert len(mine) == 1, (prompt, out[-1500:])
        assert f"Output: {tok.decode(toks)}\n" in mine[0], (prompt, mine[0], tok.decode(toks))
