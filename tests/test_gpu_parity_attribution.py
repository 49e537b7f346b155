"""Where does the full-depth distance between ours and the compiled reference come from? (VERDICT r04 item 7 — a DIAGNOSTIC,
test-harness only; runs with SWIFTLLM_PARITY_FULL_CONTROL=1, not in the driver's suite.)

32-layer Llama-3-8B (random init, the checkpoint of tests/test_gpu_parity_fulldepth.py), batch 32, 1024-token prompts, one
teacher-forced decode step. The residual stream after EVERY layer (residual + the layer's FFN output, fp32) is dumped on three
sides at identical inputs (same prompts, the reference's token fed to all):
    ref      the compiled reference under its own plan (split width from its heuristic),
    ref'     the compiled reference under another legal split width (128)  — its self-distance, a ONE-site perturbation
             (flash-decoding merge order),
    ours     the product (eager launches, default path).
Per layer l the report holds d(ours, ref)[l] and d(ref', ref)[l] in ulps of the storage dtype at the row's scale (max over
the 32 rows), their ratio, and the per-layer GROWTH d[l] / d[l-1]. How to read it: a distance that one operator site
explains would enter at one depth and then only be amplified (growth ~ that of the one-site control); a distance that
every rounding site of every layer feeds grows faster than the control at EVERY depth. The test asserts nothing about the
ratio (tests/test_gpu_parity_fulldepth.py does the bounding) — it writes gpurun_out/parity_attribution_<dtype>.json and prints
the table DESIGN.md section 6 quotes."""
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch

from oracle import synth
from tests import _parity as P

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1800),
              pytest.mark.skipif(not P.STAGED, reason="oracle/_ref not staged (python -m oracle.make_ref)"),
              pytest.mark.skipif(os.environ.get("SWIFTLLM_PARITY_FULL_CONTROL") != "1",
                                 reason="diagnostic: set SWIFTLLM_PARITY_FULL_CONTROL=1")]

PROMPT, BATCH = 1024, 32


def _row_ulps(a: torch.Tensor, b: torch.Tensor, tdtype) -> torch.Tensor:
    """max over hidden of |a - b| per row, in ulps of tdtype at the row's largest |b| -> [layers, rows]"""
    d = (a - b).abs().amax(dim=-1)
    return d / P.ulp(b.abs().amax(dim=-1), tdtype)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_per_layer_attribution_of_the_full_depth_distance(tmp_path, dtype):
    from swiftllm_amd.worker.kernels.linear import RawResidual, SplitKPartials
    from swiftllm_amd.worker.layers.transformer_layer import LlamaTransformerLayer
    tdtype = torch.bfloat16 if dtype == "bfloat16" else torch.float16
    cfg = synth.make_config(**synth.LLAMA3_8B)
    path = str(tmp_path / "llama3_8b")
    os.makedirs(path)
    sd = synth.make_state_dict_on_gpu(cfg, seed=2024, dtype=torch.float16)
    synth.write_model_dir(path, cfg, sd)
    del sd
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(77)
    prompts = [torch.randint(0, cfg["vocab_size"], (PROMPT,), generator=g).tolist() for _ in range(BATCH)]
    seq_ids = list(range(BATCH))
    script = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[]),
              dict(input_ids=None, seq_ids=seq_ids, dec_lens=[PROMPT + 1] * BATCH)]

    def reference(tag, **extra):
        job, out = tmp_path / f"{tag}_job.pt", tmp_path / f"{tag}.pt"
        torch.save(dict(config=cfg, model_path=path, num_blocks=BATCH * 68 + 4, max_len=PROMPT + 16, steps=script, dtype=dtype,
                        logits="none", dump_residual_steps=[1], **extra), job)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("TRITON_INTERPRET", None)
        r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(job), str(out)], cwd=P.ROOT, env=env,
                           capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        toks = [x["tokens"] for x in torch.load(out, weights_only=False)]
        return toks, torch.load(str(out) + ".residual.pt", weights_only=False)[1]

    ref_toks, ref_res = reference("ref")
    script[1]["input_ids"] = [[t] for t in ref_toks[0]]        # every other party is fed the reference's first token
    _, self_res = reference("ref_sbs128", seq_block_size=128)

    # ---- ours, eager, with a recording wrapper around every layer (test harness only) ----
    model = P.our_model(path, dtype, BATCH, PROMPT, 2, use_hip_graph=False)
    model.forward(prompts, seq_ids, [])
    log = []
    orig = LlamaTransformerLayer.forward

    def recording(self, input_embds, residual_buf, *a, **kw):
        out = orig(self, input_embds, residual_buf, *a, **kw)
        if isinstance(out, RawResidual):
            stream = residual_buf.float()
        elif isinstance(out, SplitKPartials):
            stream = residual_buf.float() + out.materialize().float()
        else:
            stream = residual_buf.float() + out.float()
        log.append(stream.cpu())
        return out
    LlamaTransformerLayer.forward = recording
    try:
        model.forward([[t] for t in ref_toks[0]], seq_ids, [PROMPT + 1] * BATCH)
    finally:
        LlamaTransformerLayer.forward = orig
    ours_res = torch.stack(log)
    del model
    torch.cuda.empty_cache()
    shutil.rmtree(path, ignore_errors=True)

    assert ours_res.shape == ref_res.shape == self_res.shape, (ours_res.shape, ref_res.shape, self_res.shape)
    d_ours = _row_ulps(ours_res, ref_res, tdtype).amax(dim=1)
    d_self = _row_ulps(self_res, ref_res, tdtype).amax(dim=1)
    rows = []
    for l in range(d_ours.numel()):
        rows.append(dict(layer=l, ours_vs_ref_ulp=round(float(d_ours[l]), 3), ref_self_ulp=round(float(d_self[l]), 3),
                         ratio=round(float(d_ours[l] / d_self[l].clamp(min=1e-9)), 3),
                         ours_growth=round(float(d_ours[l] / d_ours[l - 1].clamp(min=1e-9)), 3) if l else None,
                         self_growth=round(float(d_self[l] / d_self[l - 1].clamp(min=1e-9)), 3) if l else None))
    first_nonzero_self = next((r["layer"] for r in rows if r["ref_self_ulp"] > 0), None)
    report = dict(dtype=dtype, batch=BATCH, prompt_len=PROMPT, step="first decode step, teacher-forced",
                  what="residual stream after each layer; max over rows of max|a-b| in ulps of the row scale",
                  per_layer=rows, first_layer_where_the_reference_differs_from_itself=first_nonzero_self,
                  final=dict(ours_vs_ref_ulp=rows[-1]["ours_vs_ref_ulp"], ref_self_ulp=rows[-1]["ref_self_ulp"],
                             ratio=rows[-1]["ratio"]))
    P.write_report(f"parity_attribution_{dtype}.json", report)
    print("\n[attribution]", dtype, "layer: ours/ref' ulps (ratio)",
          "  ".join(f"{r['layer']}:{r['ours_vs_ref_ulp']}/{r['ref_self_ulp']}({r['ratio']})" for r in rows[::4] + [rows[-1]]))
