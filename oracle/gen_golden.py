"""oracle/gen_golden.py — freeze outputs of the REFERENCE ITSELF as golden vectors (TEST INFRASTRUCTURE).

Runs only in the build container, where /root/reference is mounted:

    python -m oracle.gen_golden            # rewrites tests/golden/*.pt

It imports the reference package unmodified and executes its own Triton kernels through Triton's CPU
interpreter (TRITON_INTERPRET=1), then its whole `LlamaModel.forward` on a tiny random model, and
stores inputs + outputs as small .pt fixtures. The harness below only makes the reference importable
and CPU-runnable; it does not change any arithmetic:
  * stub modules for the three imports that are not installed (`ray`, `vllm_flash_attn`,
    `swiftllm_c`); the vllm_flash_attn stub forwards to the reference's own `prefill_attention`
    (the drop-in the reference itself shows commented out at transformer_layer.py:97-100);
  * wrappers that make torch's factory functions read `device="cuda"` as `device="cpu"`, and no-op
    stand-ins for torch.cuda streams/events.
The fixtures travel to the GPU box; /root/reference does not.
"""
import contextlib
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"

import torch  # noqa: E402

REFERENCE = "/root/reference"
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


_FACTORIES = ("tensor", "zeros", "ones", "empty", "arange", "full")


class _CudaToCpu:
    """Context manager: while active, torch's factory functions treat device="cuda" as "cpu" (the
    reference hard-codes device="cuda" in every constructor, e.g. model.py:272, block_manager.py:29).
    Plain function wrappers on the torch module — a TorchFunctionMode also intercepts the Triton
    interpreter's own internals, which crashes it."""

    def __enter__(self):
        self._saved = {n: getattr(torch, n) for n in _FACTORIES}
        for name, fn in self._saved.items():
            def wrapped(*args, __fn=fn, **kwargs):
                dev = kwargs.get("device")
                if dev is not None and str(dev).startswith("cuda"):
                    kwargs["device"] = "cpu"
                return __fn(*args, **kwargs)
            setattr(torch, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(torch, name, fn)
        return False


class _NoStream:
    def wait_event(self, *_):
        pass

    def wait_stream(self, *_):
        pass


class _NoEvent:
    def record(self, *_):
        pass


def _install_harness():
    for name in ("ray", "vllm_flash_attn", "swiftllm_c"):
        m = types.ModuleType(name)
        if name == "ray":
            m.remote = lambda cls: cls
        sys.modules[name] = m
    sys.path.insert(0, REFERENCE)
    torch.cuda.Stream = _NoStream
    torch.cuda.Event = _NoEvent
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *a: _NoStream()
    torch.cuda.default_stream = lambda *a: _NoStream()
    torch.cuda.get_device_name = lambda *a, **k: "cpu"

    import swiftllm  # noqa: F401  (the reference package)
    from swiftllm.worker.kernels.prefill_attn import prefill_attention as ref_prefill

    def flash_attn_varlen_func(q, k, v, cu_q, cu_k, max_q, max_k, softmax_scale=None, causal=True):
        assert causal
        # Reference quirk: model.py:340-343 terminates prefill_seq_start_locs_with_end with
        # num_tokens (prefill + decode) instead of num_prefill_tokens, so on a piggybacked batch the
        # last prefill sequence appears to extend over the decoding tokens and the attention call
        # reads/writes past q[:P] (undefined behaviour; the reference's scheduler never emits mixed
        # batches, scheduler.py:93-94, so it never trips). The intended value is the row count of q.
        cu_q = cu_q.clone()
        cu_q[-1] = min(int(cu_q[-1]), q.shape[0])
        st = types.SimpleNamespace(
            num_prefill_seqs=cu_q.numel() - 1, max_prefill_len=max_q, softmax_scale=softmax_scale,
            prefill_seq_start_locs=cu_q[:-1].contiguous(),
            prefill_seq_lens=(cu_q[1:] - cu_q[:-1]).contiguous())
        mc = types.SimpleNamespace(num_q_heads=q.shape[1], num_kv_heads=k.shape[1], head_dim=q.shape[2])
        o = torch.zeros_like(q)
        ref_prefill(q.contiguous(), k.contiguous(), v.contiguous(), o, mc, None, st)
        return o

    sys.modules["vllm_flash_attn"].flash_attn_varlen_func = flash_attn_varlen_func


def _save(name, obj):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name)
    torch.save(obj, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def _mk_state(**kw):
    return types.SimpleNamespace(**kw)


def gen_elementwise():
    from swiftllm.worker.kernels.rmsnorm import rmsnorm_inplace, fused_add_rmsnorm_inplace
    from swiftllm.worker.kernels.silu_and_mul import silu_and_mul_inplace
    from swiftllm.worker.kernels.rotary_emb import rotary_embedding_inplace
    g = torch.Generator().manual_seed(11)
    out = {}
    # rmsnorm / fused add rmsnorm: hidden must be a power of two for tl.arange
    x = torch.randn(7, 256, generator=g).half()
    r = torch.randn(7, 256, generator=g).half()
    w = (1 + 0.1 * torch.randn(256, generator=g)).half()
    x1 = x.clone()
    rmsnorm_inplace(x1, w, 1e-5)
    x2, r2 = x.clone(), r.clone()
    fused_add_rmsnorm_inplace(x2, r2, w, 1e-5)
    out["rmsnorm"] = dict(x=x, w=w, eps=1e-5, out=x1)
    out["fused_add_rmsnorm"] = dict(x=x, r=r, w=w, eps=1e-5, out_x=x2, out_r=r2)
    # silu
    xs = (2 * torch.randn(5, 1024, generator=g)).half()
    xs1 = xs.clone()
    silu_and_mul_inplace(xs1)
    out["silu_and_mul"] = dict(x=xs, out=xs1)
    # rotary: T=6, H=8, KVH=2, D=64
    q = torch.randn(6, 8, 64, generator=g).half()
    k = torch.randn(6, 2, 64, generator=g).half()
    ang = torch.rand(6, 32, generator=g) * 6.28
    cos, sin = torch.cos(ang).half(), torch.sin(ang).half()
    q1, k1 = q.clone(), k.clone()
    rotary_embedding_inplace(q1, k1, _mk_state(position_cos=cos, position_sin=sin))
    out["rotary"] = dict(q=q, k=k, cos=cos, sin=sin, out_q=q1, out_k=k1)
    _save("elementwise.pt", out)


from .synth import paged_setup as _paged_setup, seeded_paged_case  # noqa: E402


def gen_paged_attention():
    from swiftllm.worker.kernels import paged_attn as pa
    from swiftllm.model_config import LlamaModelConfig
    from swiftllm.engine_config import EngineConfig
    cases = {}
    g = torch.Generator().manual_seed(12)
    specs = {
        # name: (H, KVH, D, lens, seq_block_size)
        "gqa4_d128": (8, 2, 128, [1, 37, 150, 64], 64),
        "mha_d64": (4, 4, 64, [17, 200], 128),
        "gqa2_d32": (4, 2, 32, [5, 48, 131], 64),
        "llama3_heads": (32, 8, 128, [150, 77], 64),
    }
    for name, (H, KVH, D, lens, sbs) in specs.items():
        L, bs = (1, 16) if name == "llama3_heads" else (2, 16)     # keep the fixture small
        layer = L - 1
        seq_ids = list(range(1, 1 + len(lens)))
        nblk = sum((n + bs - 1) // bs for n in lens) + 3
        k_cache, v_cache, bt = _paged_setup(g, nblk, L, KVH, bs, D, seq_ids, lens)
        q = torch.randn(len(lens), H, D, generator=g).half()
        mc = LlamaModelConfig(dict(model_type="llama", hidden_act="silu", num_hidden_layers=L,
                                   num_attention_heads=H, num_key_value_heads=KVH, hidden_size=H * D,
                                   vocab_size=8, max_position_embeddings=512, intermediate_size=16,
                                   rms_norm_eps=1e-5))
        ec = EngineConfig(model_path="", use_dummy=True, block_size=bs, gpu_mem_utilization=0.9,
                          num_cpu_blocks=0, max_seqs_in_block_table=bt.shape[0],
                          max_blocks_per_seq=bt.shape[1], max_batch_size=8, max_tokens_in_batch=1024)
        nsb = (max(lens) + sbs - 1) // sbs
        st = _mk_state(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs,
                       num_seq_blocks=nsb, softmax_scale=D ** -0.5,
                       decoding_seq_lens=torch.tensor(lens, dtype=torch.int32),
                       seq_ids=torch.tensor(seq_ids, dtype=torch.int32))
        o = torch.zeros(len(lens), H, D, dtype=torch.float16)
        # capture the reference's partials as well: re-run phase 1 with our own buffers
        mid_o = torch.zeros(len(lens), H, nsb, D, dtype=torch.float32)
        mid_lse = torch.full((len(lens), H, nsb), float("-inf"), dtype=torch.float32)
        pa._fwd_paged_attention_phase1[(len(lens), H, nsb)](
            mid_o, mid_lse, q, k_cache, v_cache, bt, st.softmax_scale * 1.442695040888963,
            st.decoding_seq_lens, st.seq_ids, nsb, layer, L, H, KVH, H // KVH, bs, D, sbs, bt.shape[1],
            num_warps=1, num_stages=4)
        pa.paged_attention(q, k_cache, v_cache, bt, mc, ec, st, layer, o)
        cases[name] = dict(H=H, KVH=KVH, D=D, L=L, block_size=bs, layer=layer, lens=lens,
                           seq_ids=seq_ids, seq_block_size=sbs, q=q, k_cache=k_cache,
                           v_cache=v_cache, block_table=bt, out=o, mid_o=mid_o, mid_lse=mid_lse)
        print("paged", name, "done")
    _save("paged_attention.pt", cases)


def gen_paged_attention_real_geometry():
    """Llama-3-8B head geometry at a configs[2]-sized context: 32 q / 8 kv heads of 128, contexts 1100 and 1024,
    seq_block_size 256 (the reference's own choice at batch 32 x ~1k, model.py:305-324) -> 5 seq-blocks, phase 2
    included. ~30 s in the interpreter. Inputs are NOT stored (9 MB): seed + checksum + outputs."""
    from swiftllm.worker.kernels import paged_attn as pa
    from swiftllm.model_config import LlamaModelConfig
    from swiftllm.engine_config import EngineConfig
    H, KVH, D, L, lens, sbs, seed, bs = 32, 8, 128, 1, [1100, 1024], 256, 1234, 16
    seq_ids, k_cache, v_cache, bt, q, checksum = seeded_paged_case(seed, H, KVH, D, L, lens)
    mc = LlamaModelConfig(dict(model_type="llama", hidden_act="silu", num_hidden_layers=L, num_attention_heads=H,
                               num_key_value_heads=KVH, hidden_size=H * D, vocab_size=8, max_position_embeddings=2048,
                               intermediate_size=16, rms_norm_eps=1e-5))
    ec = EngineConfig(model_path="", use_dummy=True, block_size=bs, gpu_mem_utilization=0.9, num_cpu_blocks=0,
                      max_seqs_in_block_table=bt.shape[0], max_blocks_per_seq=bt.shape[1], max_batch_size=8,
                      max_tokens_in_batch=4096)
    nsb = (max(lens) + sbs - 1) // sbs
    st = _mk_state(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
                   softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32),
                   seq_ids=torch.tensor(seq_ids, dtype=torch.int32))
    o = torch.zeros(len(lens), H, D, dtype=torch.float16)
    pa.paged_attention(q, k_cache, v_cache, bt, mc, ec, st, 0, o)
    _save("paged_attention_llama3_1k.pt", dict(H=H, KVH=KVH, D=D, L=L, block_size=bs, layer=0, lens=lens, seq_ids=seq_ids,
                                                seq_block_size=sbs, seed=seed, kv_checksum=checksum, q=q, out=o))


def gen_prefill_attention():
    from swiftllm.worker.kernels.prefill_attn import prefill_attention
    g = torch.Generator().manual_seed(13)
    cases = {}
    specs = {"gqa2_d64": (4, 2, 64, [5, 40, 130]), "gqa4_d128": (8, 2, 128, [33, 129]),
             "mha_d32": (4, 4, 32, [17, 64, 1])}
    for name, (H, KVH, D, lens) in specs.items():
        P = sum(lens)
        q = torch.randn(P, H, D, generator=g).half()
        k = torch.randn(P, KVH, D, generator=g).half()
        v = torch.randn(P, KVH, D, generator=g).half()
        o = torch.zeros(P, H, D, dtype=torch.float16)
        lens_t = torch.tensor(lens, dtype=torch.int32)
        starts = torch.cumsum(lens_t, 0, dtype=torch.int32) - lens_t
        st = _mk_state(num_prefill_seqs=len(lens), max_prefill_len=max(lens), softmax_scale=D ** -0.5,
                       prefill_seq_start_locs=starts, prefill_seq_lens=lens_t)
        mc = _mk_state(num_q_heads=H, num_kv_heads=KVH, head_dim=D)
        prefill_attention(q, k, v, o, mc, None, st)
        cases[name] = dict(H=H, KVH=KVH, D=D, lens=lens, q=q, k=k, v=v, out=o)
        print("prefill", name, "done")
    _save("prefill_attention.pt", cases)


def gen_kvcache_and_blocks():
    from swiftllm.worker.kernels.kvcache_mgmt import store_kvcache
    from swiftllm.worker.block_manager import BlockManager
    g = torch.Generator().manual_seed(14)
    out = {}
    # ---- store_kvcache on a mixed batch: 2 prefill seqs (lens 21, 16) + 2 decoding seqs ------------
    L, KVH, bs, D = 3, 2, 16, 32
    max_seqs, mbps, num_blocks, layer = 8, 16, 14, 2
    with _CudaToCpu():
        mgr = BlockManager("GPU", num_blocks, max_seqs, mbps, bs)
        mgr.block_table.zero_()
        # decoding seqs 5 and 2 already own blocks for lengths 17 and 40 (before this step's token)
        mgr.allocate_blocks_for_seqs(torch.tensor([5, 2], dtype=torch.int32), torch.tensor([17, 40], dtype=torch.int32))
        seq_ids = torch.tensor([3, 0, 5, 2], dtype=torch.int32)
        plens = [21, 16]
        dlens = [18, 49]        # 49 needs a 4th block for seq 2
        seq_lengths = torch.tensor(plens + dlens, dtype=torch.int32)
        new_blocks = mgr.allocate_blocks_for_seqs(seq_ids, seq_lengths)
    T = sum(plens) + 2
    k = torch.randn(T, KVH, D, generator=g).half()
    v = torch.randn(T, KVH, D, generator=g).half()
    k_cache = torch.zeros(num_blocks, L, KVH, bs, D, dtype=torch.float16)
    v_cache = torch.zeros_like(k_cache)
    plens_t = torch.tensor(plens, dtype=torch.int32)
    st = _mk_state(seq_ids=seq_ids, num_prefill_seqs=2, num_prefill_tokens=sum(plens),
                   max_prefill_len=max(plens), prefill_seq_lens=plens_t,
                   prefill_seq_start_locs=torch.cumsum(plens_t, 0, dtype=torch.int32) - plens_t,
                   num_decoding_seqs=2, decoding_seq_lens=torch.tensor(dlens, dtype=torch.int32))
    mc = _mk_state(num_layers=L, num_kv_heads=KVH, head_dim=D)
    ec = _mk_state(block_size=bs, max_blocks_per_seq=mbps)
    store_kvcache(k, v, k_cache, v_cache, mgr.block_table, mc, ec, st, layer)
    out["store_kvcache"] = dict(L=L, KVH=KVH, block_size=bs, D=D, layer=layer, seq_ids=seq_ids,
                                plens=plens, dlens=dlens, k=k, v=v,
                                block_table=mgr.block_table.clone(), k_cache=k_cache, v_cache=v_cache,
                                new_blocks=new_blocks.clone())
    # ---- a scripted life of a BlockManager: allocate / grow / free / gather, state after each step ---
    with _CudaToCpu():
        mgr = BlockManager("GPU", 24, 6, 8, bs)
        mgr.block_table.zero_()
        script, trace = [
            ("alloc", [0, 1, 2], [40, 16, 1]),
            ("alloc", [1, 2, 4], [17, 33, 100]),
            ("free", [1], None),
            ("alloc", [3, 0], [50, 49]),
            ("gather", [4, 2], None),
            ("alloc", [5, 1], [20, 70]),
            ("free", [0, 3, 5, 1], None),
        ], []
        for op, ids, lens in script:
            ids_t = torch.tensor(ids, dtype=torch.int32)
            ret = None
            if op == "alloc":
                ret = mgr.allocate_blocks_for_seqs(ids_t, torch.tensor(lens, dtype=torch.int32)).clone()
            elif op == "free":
                mgr.free_blocks_for_seqs(ids_t)
            else:
                ret = mgr.gather_allocated_blocks_and_free(ids_t).clone()
            n = mgr.num_seq_allocated_blocks.clone()
            bt = mgr.block_table.clone()
            for s in range(bt.shape[0]):
                bt[s, int(n[s]):] = -1          # entries past the count are don't-care
            trace.append(dict(op=op, ids=ids, lens=lens, ret=ret, num_free=mgr.num_free_blocks,
                              num_alloc=n, block_table=bt, is_free=mgr.is_block_free.clone()))
    out["block_manager_trace"] = dict(num_blocks=24, max_seqs=6, mbps=8, block_size=bs, trace=trace)
    _save("kvcache_blocks.pt", out)


def gen_rope_tables():
    from swiftllm.worker.model import LlamaModel
    out = {}
    for name, scaling in (("scalar1", None), ("scalar4", 4.0),
                          ("dict", dict(factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                        original_max_position_embeddings=256, rope_type="llama3"))):
        mc = _mk_state(rope_scaling=1.0 if scaling is None else scaling, rope_theta=500000.0,
                       max_position_embeddings=256, head_dim=64)
        fake = _mk_state(model_config=mc)
        with _CudaToCpu():
            LlamaModel._init_to_get_rotary(fake)
        rows = torch.tensor([0, 1, 2, 17, 100, 255, fake._cos_cached.shape[0] - 1])
        out[name] = dict(rope_scaling=scaling, rope_theta=500000.0, max_position_embeddings=256,
                         head_dim=64, num_rows=fake._cos_cached.shape[0], rows=rows,
                         cos=fake._cos_cached[rows].clone(), sin=fake._sin_cached[rows].clone())
    _save("rope_tables.pt", out)


def gen_e2e():
    """The reference's whole LlamaModel.forward (fp16, Triton interpreter) on BASELINE configs[0]'s
    model: 3 prompts prefill, 6 decode steps, then a mixed (piggybacked) batch; token ids and
    pre-argmax logits of every step."""
    import swiftllm
    from swiftllm.worker.weight import LlamaWeight
    from swiftllm.worker.layers.pre_layer import LlamaPreLayer
    from swiftllm.worker.layers.transformer_layer import LlamaTransformerLayer
    from swiftllm.worker.layers import post_layer as post_mod
    from . import synth
    import tempfile

    cfg = synth.make_config()
    sd = synth.make_state_dict(cfg, seed=0, dtype=torch.float16)
    tmp = tempfile.mkdtemp()
    synth.write_model_dir(tmp, cfg)
    ec = swiftllm.EngineConfig(model_path=tmp, use_dummy=False, block_size=16, gpu_mem_utilization=0.9,
                               num_cpu_blocks=8, max_seqs_in_block_table=16, max_blocks_per_seq=32,
                               max_batch_size=8, max_tokens_in_batch=256)
    logits_log = []
    orig_linear = post_mod.linear

    def tapped_linear(a, w):
        r = orig_linear(a, w)
        logits_log.append(r.float().clone())
        return r
    post_mod.linear = tapped_linear     # the only linear() in post_layer is lm_head (post_layer.py:38)

    with _CudaToCpu():
        model = swiftllm.LlamaModel(ec)
        weight = LlamaWeight(model.model_config, torch.float16)
        for item in weight.registered_weights:
            setattr(weight, item.attr_name, sd[item.key].clone())
        for layer in weight.layers:
            for item in layer.registered_weights:
                setattr(layer, item.attr_name, sd[item.key].clone())
            layer.up_gate_proj = torch.cat((layer.up_proj, layer.gate_proj), dim=0).contiguous()
        model.weight = weight
        model._init_to_get_rotary()
        model.pre_layer = LlamaPreLayer(model.model_config, weight)
        model.transformer_layers = [
            LlamaTransformerLayer(model.model_config, ec, weight.layers[i], _NoStream(), i)
            for i in range(model.model_config.num_layers)]
        model.post_layer = post_mod.LlamaPostLayer(model.model_config, weight)
        model.init_kvcache_and_swap(24)
        with torch.inference_mode():
            model.gpu_block_manager.block_table.zero_()

        g = torch.Generator().manual_seed(1)
        prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in (5, 9, 17)]
        steps = []
        toks = model.forward(prompts, [0, 1, 2], [])
        steps.append(dict(kind="prefill", input_ids=prompts, seq_ids=[0, 1, 2], dec_lens=[],
                          tokens=toks, logits=logits_log[-1]))
        lens = [len(p) for p in prompts]
        last = toks
        for _ in range(6):
            lens = [n + 1 for n in lens]
            toks = model.forward([[t] for t in last], [0, 1, 2], list(lens))
            steps.append(dict(kind="decode", input_ids=[[t] for t in last], seq_ids=[0, 1, 2],
                              dec_lens=list(lens), tokens=toks, logits=logits_log[-1]))
            last = toks
        # piggybacked step: a new 20-token prompt (seq 3) together with the 3 decoding sequences
        new_prompt = torch.randint(0, cfg["vocab_size"], (20,), generator=g).tolist()
        lens = [n + 1 for n in lens]
        ids = [new_prompt] + [[t] for t in last]
        toks = model.forward(ids, [3, 0, 1, 2], list(lens))
        steps.append(dict(kind="mixed", input_ids=ids, seq_ids=[3, 0, 1, 2], dec_lens=list(lens),
                          tokens=toks, logits=logits_log[-1]))
    post_mod.linear = orig_linear
    _save("e2e_tiny_fp16.pt", dict(config=cfg, seed=0, engine=dict(
        block_size=16, num_cpu_blocks=8, max_seqs_in_block_table=16, max_blocks_per_seq=32,
        max_batch_size=8, max_tokens_in_batch=256, num_gpu_blocks=24), steps=steps))


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit(f"{REFERENCE} is not mounted: golden vectors can only be regenerated in the "
                         "build container")
    _install_harness()
    only = set(sys.argv[1:])
    for fn in (gen_elementwise, gen_kvcache_and_blocks, gen_rope_tables, gen_prefill_attention,
               gen_paged_attention, gen_paged_attention_real_geometry, gen_e2e):
        if not only or fn.__name__ in only:
            fn()


if __name__ == "__main__":
    main()
