"""End-to-end parity at Llama-3-8B WIDTH against the independent checker.

Model: Llama-3-8B layer geometry (hidden 4096, 32 q / 8 kv heads of 128, FFN 14336, rope_theta 5e5), 2 layers,
8k vocabulary (so the CPU side stays small). Workloads: BASELINE.json configs[1] (batch 1, 1024-token prompt,
8 greedy steps) and a configs[2]-shaped ragged batch of 32 (lengths 1..1024, 3 steps; r02-r03 ran 12 / 5 steps: the
per-step distances do not grow with the step, profiles/r02_parity_fullwidth_*). The HIP data plane — default path,
hipGraph replay, row-major (unpacked) decode weights — is compared with oracle/ref_model.py, which is pinned to the
reference's own run on the tiny golden (tests/test_oracle_golden.py), in BOTH decode-score modes (the default path
defers the RMSNorm scale into the consuming projection, `exact_rmsnorm_rounding` keeps the reference's rounding points):
  * "fp32": exact scores (the reference's commented eager restatement, paged_attn.py:224-259);
  * "ref" : the Triton kernel's fp16 products / fp16 sum / fp16 scale (paged_attn.py:17,72-73).
The distance between those two oracles is the noise floor of the reference itself at this size: it is measured
and reported beside ours, and the bars below are stated against it.

Bars (fp16 — the reference's precision; written here, judged here):
  * greedy token ids: identical to the fp32-score oracle at every step and sequence, except positions where the
    oracle's own top-2 logit gap is below the measured logit distance (a near-tie no implementation can pin);
  * pre-argmax logits: max |ours - oracle_fp32| <= 3 ulp of the storage dtype at the row's scale
    (ulp(max|logit| of the row)) AND no farther from the exact oracle than the reference's own score rounding is
    (measured r02: ours 2.0 ulp, the reference 4.75-5.5 ulp); for reference, fp16 logits of magnitude 2..4 are spaced 1.95e-3 apart, so the
    north star's absolute 1e-3 is sub-ulp here and is reported, not asserted, at this width. It IS asserted
    where logits are small enough for it to be meaningful: tests/test_gpu_model.py (tiny golden).
bf16 (headline dtype; the reference has no bf16 path): same checks against the oracle run in bf16, with the
measured noise floor printed — the same 3-ulp / below-the-reference's-noise bar in bf16 ulps.
A JSON report goes to gpurun_out/parity_fullwidth_<case>.json (copied to profiles/ when committed).
"""
import json
import os

import pytest
import torch

from oracle import synth
from oracle.ref_model import RefLlamaModel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(num_hidden_layers=2, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
           intermediate_size=14336, vocab_size=8192, max_position_embeddings=2048, rope_theta=500000.0)
CASES = {
    # name: (prompt lengths, decode steps)
    "configs1_batch1": ([1024], 8),
    "configs2_batch32": ([1024, 1, 15, 16, 17, 100, 257, 640, 33, 1000, 511, 512, 513, 64, 128, 900,
                          1024, 2, 31, 48, 300, 700, 800, 5, 1023, 256, 255, 77, 450, 999, 10, 129], 3),
}
VARIANTS = (("default", dict()), ("eager_launches", dict(use_hip_graph=False)),
            ("row_major_weights", dict(pack_decode_weights=False)),
            ("exact_rmsnorm_rounding", dict(tuning=dict(defer_rmsnorm=False))),
            # batch 1 only: the whole transformer stack of the step as ONE persistent launch (csrc/decode_engine.hip), at the
            # real Llama-3-8B width, in both dtypes (it keeps the reference's rounding points in float16 too)
            ("decode_engine", dict(tuning=dict(decode_engine=True))))


def _ulp(x: torch.Tensor, dtype) -> torch.Tensor:
    """Spacing of `dtype` at magnitude |x| (fp32 tensor in, fp32 out)."""
    mant = 10 if dtype == torch.float16 else 7
    e = torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14)))
    return torch.exp2(e - mant)


def _engine_kw(batch, dtype):
    return dict(use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0,
                max_seqs_in_block_table=max(8, batch), max_blocks_per_seq=72, max_batch_size=batch,
                max_tokens_in_batch=batch * 1040, dtype=dtype)


FULL_CONTROL = os.environ.get("SWIFTLLM_PARITY_FULL_CONTROL") == "1"
# The driver's GPU tier has 1 200 s for the whole suite: float16 at batch 32 here (47 s) and the float16 mixed step (36 s) run
# with SWIFTLLM_PARITY_FULL_CONTROL=1 only since r05 — float16 at batch 32 / this geometry stays in the suite through
# test_full_width_logits_hold_the_absolute_1e3_bar_when_it_is_meaningful (oracle AND compiled reference) and the decisive test.
FORWARD_CASES = [(c, d) for c in sorted(CASES) for d in ("float16", "bfloat16")
                 if FULL_CONTROL or not (c == "configs2_batch32" and d == "float16")]


@pytest.mark.parametrize("case,dtype", FORWARD_CASES)
def test_full_width_forward_matches_oracle(tmp_path, case, dtype):
    from swiftllm_amd import EngineConfig, LlamaModel, LlamaModelConfig
    lens, steps = CASES[case]
    batch = len(lens)
    tdtype = torch.float16 if dtype == "float16" else torch.bfloat16
    cfg = synth.make_config(**CFG)
    sd = synth.make_state_dict(cfg, seed=31, dtype=tdtype)
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in lens]
    seq_ids = list(range(batch))
    num_blocks = sum(-(-(n + steps + 1) // 16) for n in lens) + 4
    kw = _engine_kw(batch, dtype)

    # ---- the checker: oracle with exact scores (teacher), and with the reference kernel's score rounding ----
    # (one prompt pass: it does not depend on the decode-score rounding; the second oracle forks from its state)
    def continue_oracle(ref, first_toks, forced=None):
        toks, logits = [first_toks], [ref.last_logits.clone()]
        cur = list(lens)
        for s in range(steps):
            cur = [n + 1 for n in cur]
            feed = forced[s] if forced is not None else toks[-1]
            toks.append(ref.forward([[t] for t in feed], seq_ids, list(cur)))
            logits.append(ref.last_logits.clone())
        return toks, logits

    teacher = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(model_path="", **kw), sd, tdtype, score_dtype="fp32")
    teacher.init_kvcache_and_swap(num_blocks)
    first = teacher.forward(prompts, seq_ids, [])
    noisy = teacher.fork(score_dtype="ref")
    want_toks, want_logits = continue_oracle(teacher, first)
    noise_toks, noise_logits = continue_oracle(noisy, first, forced=want_toks)
    del teacher, noisy

    # ---- the product ---------------------------------------------------------------------------------------
    synth.write_model_dir(str(tmp_path), cfg, sd)

    def run_hip(opts):
        model = LlamaModel(EngineConfig(model_path=str(tmp_path), **kw, **opts))
        model.load_weights()
        model.init_kvcache_and_swap(num_blocks)
        model.post_layer.logits_tap = []
        tap = model.post_layer.logits_tap
        toks, logits = [model.forward(prompts, seq_ids, [])], [tap[-1].float().cpu()]
        cur = list(lens)
        for s in range(steps):
            cur = [n + 1 for n in cur]
            toks.append(model.forward([[t] for t in want_toks[s]], seq_ids, list(cur)))   # teacher-forced
            logits.append(tap[-1].float().cpu())
        del model
        torch.cuda.empty_cache()
        return toks, logits

    def compare(toks, logits):
        """per step: max |dlogit|, the same in ulps of the row scale, token mismatches and whether each mismatch
        sits on a near-tie of the teacher"""
        rows = []
        for s, (a, b) in enumerate(zip(logits, want_logits)):
            d = (a - b).abs()
            row_ulp = _ulp(b.abs().amax(dim=1, keepdim=True), tdtype)
            mism = [i for i, (x, y) in enumerate(zip(toks[s], want_toks[s])) if x != y]
            ties = []
            for i in mism:
                top2 = b[i].topk(2).values
                ties.append(float(top2[0] - top2[1]))
            rows.append(dict(step=s, max_abs=float(d.max()), max_ulp_of_row=float((d / row_ulp).max()),
                             mismatches=len(mism), mismatch_top2_gaps=ties,
                             mismatch_row_max_abs=[float(d[i].max()) for i in mism]))
        return rows

    report = dict(case=case, dtype=dtype, batch=batch, decode_steps=steps, model=CFG,
                  noise_floor_ref_scores_vs_exact=compare(noise_toks, noise_logits))
    noise_ulp = max(r["max_ulp_of_row"] for r in report["noise_floor_ref_scores_vs_exact"])
    failures = []
    for name, opts in VARIANTS:
        if batch > 1 and name in ("eager_launches", "row_major_weights", "decode_engine"):
            continue        # (suite time: both run at batch 1 here; graph == eager bit-equality at batch 32 is tests/test_gpu_model.py's)
        toks, logits = run_hip(opts)
        rows = compare(toks, logits)
        report[name] = rows
        worst_abs = max(r["max_abs"] for r in rows)
        worst_ulp = max(r["max_ulp_of_row"] for r in rows)
        if worst_ulp > 3.0 or worst_ulp > noise_ulp:
            failures.append(f"{name}: logits off by {worst_ulp:.2f} ulp of the row scale ({worst_abs:.2e} abs); "
                            f"the reference's own score rounding is {noise_ulp:.2f} ulp from the exact oracle")
        for r in rows:
            # PER ROW: a greedy id may differ from the oracle's only where the oracle's own top-2 gap in
            # that row is within twice THAT ROW's logit distance — not the run's worst
            for gap, row_abs in zip(r["mismatch_top2_gaps"], r["mismatch_row_max_abs"]):
                if gap > 2 * row_abs:
                    failures.append(f"{name}: token mismatch at step {r['step']} with oracle top-2 gap {gap:.2e} "
                                    f"> 2 x the row's logit distance {row_abs:.2e}")
    def summarise(rows):
        return dict(max_abs=max(r["max_abs"] for r in rows), max_ulp_of_row=max(r["max_ulp_of_row"] for r in rows),
                    token_mismatches=sum(r["mismatches"] for r in rows),
                    tokens_compared=(steps + 1) * batch)
    report["summary"] = {k: summarise(v) for k, v in report.items() if isinstance(v, list)}
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fullwidth_{case}_{dtype}.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    print("\n[full-width parity]", case, dtype, json.dumps(report["summary"]))
    assert not failures, failures


STAGED = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "swiftllm", "worker", "model.py"))


@pytest.mark.parametrize("shrink", [32], ids=["logits_below_0.25"])     # (16: measured in r03, see the docstring — dividing
# an fp16 lm_head by 2 is exact, so every distance at 32 is exactly half of its value at 16: one run says both)
def test_full_width_logits_hold_the_absolute_1e3_bar_when_it_is_meaningful(tmp_path, shrink):
    """north_star: "pre-argmax logits within 1e-3 of the reference Triton path". At |logit| 4-8 that is a quarter of an
    fp16 ulp (spacing 3.9e-3) — not a bar any fp16 implementation, the reference's own two paths included, can be held
    to. Here the SAME Llama-3-8B-width model (hidden 4096, 32/8 heads of 128, FFN 14336, 2 layers, batch 32 at ~1k
    contexts, fp16) gets an lm_head drawn `shrink` times smaller so that 1e-3 is a real bound, and it is ASSERTED for
    prefill + 3 decode steps:
      * |logit| <= 0.5 (1e-3 >= 4 fp16 ulps): ours is within 1e-3 of the CPU oracle with exact scores (measured 4.9e-4).
        The compiled reference itself is 1.66e-3 from that oracle here — it rounds decode scores to fp16
        (paged_attn.py:72-73) — so ours-vs-reference (1.62e-3) is bounded by the triangle 1e-3 + the reference's own
        distance, and that is what is asserted;
      * |logit| <= 0.25 (1e-3 >= 8 fp16 ulps): all three pairwise distances are within 1e-3, the north star's bar verbatim
        against the compiled reference Triton path."""
    import subprocess
    import sys
    from swiftllm_amd import EngineConfig, LlamaModel, LlamaModelConfig
    lens, steps = CASES["configs2_batch32"]
    batch = len(lens)
    cfg = synth.make_config(**CFG)
    sd = synth.make_state_dict(cfg, seed=31, dtype=torch.float16)
    sd["lm_head.weight"] = (sd["lm_head.weight"].float() / shrink).to(torch.float16)
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in lens]
    seq_ids = list(range(batch))
    num_blocks = sum(-(-(n + steps + 1) // 16) for n in lens) + 4
    kw = _engine_kw(batch, "float16")
    synth.write_model_dir(str(tmp_path / "model"), cfg, sd)

    ref = RefLlamaModel(LlamaModelConfig(cfg), EngineConfig(model_path="", **kw), sd, torch.float16, score_dtype="fp32")
    ref.init_kvcache_and_swap(num_blocks)
    want_toks, want_logits = [ref.forward(prompts, seq_ids, [])], [ref.last_logits.clone()]
    cur = list(lens)
    script = [dict(input_ids=prompts, seq_ids=seq_ids, dec_lens=[])]
    for s in range(steps):
        cur = [n + 1 for n in cur]
        want_toks.append(ref.forward([[t] for t in want_toks[-1]], seq_ids, list(cur)))
        want_logits.append(ref.last_logits.clone())
        script.append(dict(input_ids=[[t] for t in want_toks[-2]], seq_ids=seq_ids, dec_lens=list(cur)))
    del ref, sd
    top = max(float(l.abs().max()) for l in want_logits)
    assert top <= 8.0 / shrink, top             # the premise: 1e-3 is >= 4 (8) fp16 ulps everywhere

    report = dict(model=CFG, lm_head_scale=1 / shrink, batch=batch, steps=steps + 1, max_abs_logit=top)
    tri_logits = None
    if STAGED:      # the compiled reference on the same checkpoint, teacher-forced with the oracle's tokens
        torch.save(dict(config=cfg, model_path=str(tmp_path / "model"), num_blocks=num_blocks, max_len=1040,
                        steps=script), tmp_path / "job.pt")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("TRITON_INTERPRET", None)
        r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(tmp_path / "job.pt"),
                            str(tmp_path / "ref.pt")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        tri_logits = [x["logits"] for x in torch.load(tmp_path / "ref.pt", weights_only=False)]
        report["reference_triton_vs_oracle_max_abs"] = max(float((a - b).abs().max()) for a, b in zip(tri_logits, want_logits))
    # ours vs the compiled reference: 1e-3 outright where the reference itself is within 1e-3 of the exact oracle,
    # else 1e-3 + the reference's own distance (triangle)
    tri_bar = 1e-3 + (0.0 if report.get("reference_triton_vs_oracle_max_abs", 0.0) <= 1e-3
                      else report["reference_triton_vs_oracle_max_abs"])
    if shrink >= 32 and tri_logits is not None:
        assert report["reference_triton_vs_oracle_max_abs"] <= 1e-3, report     # (so the bar below IS 1e-3 at this scale)

    failures = []
    for name, opts in (("default", dict()), ("reference_blas_calls", dict(fuse_qkv=False, use_skinny_gemm=False))):
        model = LlamaModel(EngineConfig(model_path=str(tmp_path / "model"), **kw, **opts))
        model.load_weights()
        model.init_kvcache_and_swap(num_blocks)
        model.post_layer.logits_tap = []
        tap = model.post_layer.logits_tap
        logits = []
        for s, step in enumerate(script):
            model.forward(step["input_ids"], step["seq_ids"], step["dec_lens"])
            logits.append(tap[-1].float().cpu())
        del model
        torch.cuda.empty_cache()
        vs_oracle = max(float((a - b).abs().max()) for a, b in zip(logits, want_logits))
        entry = dict(vs_oracle_max_abs=vs_oracle)
        if vs_oracle > 1e-3:
            failures.append(f"{name}: {vs_oracle:.3e} from the exact-score oracle")
        if tri_logits is not None:
            vs_tri = max(float((a - b).abs().max()) for a, b in zip(logits, tri_logits))
            entry["vs_reference_triton_max_abs"] = vs_tri
            if vs_tri > tri_bar:
                failures.append(f"{name}: {vs_tri:.3e} from the compiled reference Triton path (bar {tri_bar:.3e})")
        report[name] = entry
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fullwidth_scaled_logits_div{shrink}_float16.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    print(f"\n[full-width, lm_head / {shrink}: the absolute 1e-3 bar]", json.dumps(report))
    assert not failures, failures


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"] if FULL_CONTROL else ["bfloat16"])
def test_piggybacked_mixed_step_at_llama3_8b_width(tmp_path, dtype):
    """VERDICT r03 item 1e — BASELINE configs[2] is "piggybacked prefill+decode (SARATHI path)": the two-stream mixed forward
    (reference transformer_layer.py:78-79,101-114) at the shape bench.py times — **4 fresh 1024-token prompts + 28 decoding
    sequences in ONE forward** at Llama-3-8B width (2 layers) — against the exact-score CPU oracle and the compiled reference.
    Script: prefill the 28 old sequences (ragged, 40..1087 tokens) -> the MIXED step -> one pure-decode step of all 32 (it
    reads the K/V the mixed step stored for both kinds of sequence). 4 x 1024 + 28 = 4124 rows: the row-block rule of
    kernels/linear.py for > 4096-row BLAS calls is on the path.
    Bars: logits within 3 ulps of the row scale of the exact oracle (decode attention as one dense fp32 softmax per sequence:
    tests/_parity.py exact_oracle); greedy ids equal except on the oracle's near-ties; and against the compiled reference:
    ours no farther from exact than it is (x 1.25 + one ulp), same greedy ids up to its near-ties. (r04 first run, with the
    block-walking oracle in both score modes, 150..1087-token sequences: ours 2.0-2.1 ulps from exact, the reference 7.5,
    the reference's score rounding alone 4.2-5.7 — profiles/r04_parity_fullwidth_mixed_step_*.json.)"""
    import subprocess
    import sys
    from swiftllm_amd import EngineConfig, LlamaModel
    from tests import _parity as P
    tdtype = torch.float16 if dtype == "float16" else torch.bfloat16
    cfg = synth.make_config(**CFG)
    sd = synth.make_state_dict(cfg, seed=33, dtype=tdtype)
    g = torch.Generator().manual_seed(9)
    old_lens = [40 + (i * 331) % 520 for i in range(28)]
    old_lens[3], old_lens[17] = 1087, 1024
    old = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in old_lens]
    new = [torch.randint(0, cfg["vocab_size"], (1024,), generator=g).tolist() for _ in range(4)]
    old_ids, new_ids = list(range(28)), [28, 29, 30, 31]
    num_blocks = sum(-(-(n + 3) // 16) for n in old_lens) + 4 * 66 + 4
    kw = dict(use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=0, max_seqs_in_block_table=32,
              max_blocks_per_seq=72, max_batch_size=32, max_tokens_in_batch=28 * 1100, dtype=dtype)
    names = ("prefill_28", "mixed_4x1024+28", "decode_32")

    def run_script(fwd, logits_of, feed=None):
        """the three forwards; `feed` = the tokens to feed (the exact oracle's), None = self-feeding"""
        out = []
        t0 = fwd(old, old_ids, [])
        out.append((t0, logits_of()))
        t1 = fwd(new + [[t] for t in (feed[0] if feed else t0)], new_ids + old_ids, [n + 1 for n in old_lens])
        out.append((t1, logits_of()))
        t2 = fwd([[t] for t in (feed[1] if feed else t1)], new_ids + old_ids, [1025] * 4 + [n + 2 for n in old_lens])
        out.append((t2, logits_of()))
        return out

    oracle = P.exact_oracle(cfg, sd, tdtype, 32, 1100)
    want = run_script(oracle.forward, lambda: oracle.last_logits.clone())
    del oracle
    feed = [want[0][0], want[1][0]]

    def distance(got):
        rows = []
        for s, ((toks, lg), (wt, wl)) in enumerate(zip(got, want)):
            d = (lg.float().cpu() - wl).abs()
            row_ulp = _ulp(wl.abs().amax(dim=1, keepdim=True), tdtype)
            ties_ok = True
            mism = [i for i, (x, y) in enumerate(zip(toks, wt)) if x != y]
            for i in mism:
                top2 = wl[i].topk(2).values
                ties_ok &= float(top2[0] - top2[1]) <= 2 * float(d[i].max())
            rows.append(dict(step=names[s], max_abs=float(d.max()), max_ulp_of_row=float((d / row_ulp).max()),
                             mismatches=len(mism), all_on_near_ties=ties_ok))
        return rows

    synth.write_model_dir(str(tmp_path / "model"), cfg, sd)
    report = dict(dtype=dtype, model=CFG, workload="28 ragged prefills -> 4 x 1024-token prompts + 28 decodes in one forward "
                  "-> 32 decodes", rows_in_mixed_step=4 * 1024 + 28)
    failures = []
    ours_logits = {}
    for name, opts in (("default", dict()),):       # (r04b also ran eager launches: bit-identical distances)
        model = LlamaModel(EngineConfig(model_path=str(tmp_path / "model"), **kw, **opts))
        model.load_weights()
        model.init_kvcache_and_swap(num_blocks)
        model.post_layer.logits_tap = []
        tap = model.post_layer.logits_tap
        got = run_script(model.forward, lambda: tap[-1].float().cpu(), feed)
        del model
        torch.cuda.empty_cache()
        report[name] = rows = distance(got)
        ours_logits[name] = [lg for _, lg in got]
        for r in rows:
            if r["max_ulp_of_row"] > 3.0:
                failures.append(f"{name} {r['step']}: {r['max_ulp_of_row']:.2f} ulp of the row scale from the exact oracle")
            if not r["all_on_near_ties"]:
                failures.append(f"{name} {r['step']}: greedy id differs from the oracle's away from a near-tie")
    if STAGED:      # the compiled reference runs the same three forwards, fed the oracle's tokens
        script = [dict(input_ids=old, seq_ids=old_ids, dec_lens=[]),
                  dict(input_ids=new + [[t] for t in feed[0]], seq_ids=new_ids + old_ids, dec_lens=[n + 1 for n in old_lens]),
                  dict(input_ids=[[t] for t in feed[1]], seq_ids=new_ids + old_ids,
                       dec_lens=[1025] * 4 + [n + 2 for n in old_lens])]
        torch.save(dict(config=cfg, model_path=str(tmp_path / "model"), num_blocks=num_blocks, max_len=1100, steps=script,
                        dtype=dtype), tmp_path / "job.pt")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("TRITON_INTERPRET", None)
        r = subprocess.run([sys.executable, "-m", "oracle.ref_triton", "forward", str(tmp_path / "job.pt"),
                            str(tmp_path / "ref.pt")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        tri = [x["logits"] for x in torch.load(tmp_path / "ref.pt", weights_only=False)]
        ref_d = [float((a - wl).abs().max()) for a, (_, wl) in zip(tri, want)]
        vs = [float((a - b).abs().max()) for a, b in zip(ours_logits["default"], tri)]
        ours_d = [r["max_abs"] for r in report["default"]]
        report["compiled_reference"] = dict(reference_vs_oracle_max_abs=ref_d, ours_vs_reference_max_abs=vs)
        # arbitration: ours is no farther from the exact oracle than the compiled reference is (x 1.25 + one ulp of the
        # largest logit), and the two pick the same greedy ids except on the reference's own near-ties
        one_ulp = [float(_ulp(wl.abs().max(), tdtype)) for _, wl in want]
        for s in range(3):
            if ours_d[s] > 1.25 * ref_d[s] + one_ulp[s]:
                failures.append(f"step {s}: ours {ours_d[s]:.3e} from exact vs the compiled reference's {ref_d[s]:.3e}")
            a, b = ours_logits["default"][s], tri[s]
            for i in (a.argmax(dim=1) != b.argmax(dim=1)).nonzero().flatten().tolist():
                top2 = b[i].topk(2).values
                if float(top2[0] - top2[1]) > 2 * float((a[i] - b[i]).abs().max()):
                    failures.append(f"step {s} seq {i}: greedy id differs from the compiled reference's away from a near-tie")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"parity_fullwidth_mixed_step_{dtype}.json"), "w", encoding="utf-8") as f:
        json.dump(report, f, indent=1)
    print("\n[mixed step at 8B width]", dtype, json.dumps(report)[:1500])
    assert not failures, failures


@pytest.mark.parametrize("name", ["scalar1", "scalar4", "dict"])
def test_product_rope_tables_match_reference_golden(tmp_path, golden, name):
    """a10: the PRODUCT's device-evaluated cos/sin tables (LlamaModel._init_to_get_rotary), incl. scalar > 1 and
    dict-style rope_scaling, against rows frozen from the reference's own _init_to_get_rotary (model.py:177-225).
    Device trig (fp32) may differ from the CPU's by an fp32 ulp before the fp16 rounding: <= 1 fp16 ulp, and the
    table length must be identical."""
    from tests.conftest import ulp_diff_fp16
    from swiftllm_amd import EngineConfig, LlamaModel
    g = golden("rope_tables.pt")[name]
    cfg = synth.make_config(num_hidden_layers=1, hidden_size=4 * g["head_dim"], num_attention_heads=4,
                            num_key_value_heads=2, intermediate_size=256, vocab_size=64,
                            max_position_embeddings=g["max_position_embeddings"], rope_theta=g["rope_theta"],
                            rope_scaling=g["rope_scaling"])
    synth.write_model_dir(str(tmp_path), cfg)
    model = LlamaModel(EngineConfig(model_path=str(tmp_path), use_dummy=True, block_size=16, gpu_mem_utilization=0.5,
                                    num_cpu_blocks=0, max_seqs_in_block_table=4, max_blocks_per_seq=8,
                                    max_batch_size=2, max_tokens_in_batch=64))
    model.load_weights()
    assert model._cos_cached.shape == (g["num_rows"], g["head_dim"] // 2)
    assert model._sin_cached.shape == (g["num_rows"], g["head_dim"] // 2)
    rows = g["rows"].long()
    cos = model._cos_cached[rows.cuda()].cpu()
    sin = model._sin_cached[rows.cuda()].cpu()
    assert cos.dtype == torch.float16
    # |x| near 0 crosses many fp16 binades: compare in absolute terms there, in ulps elsewhere
    for got, want in ((cos, g["cos"]), (sin, g["sin"])):
        big = want.abs() >= 2.0 ** -6
        assert ulp_diff_fp16(torch.where(big, got, want), want) <= 1
        assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -11
