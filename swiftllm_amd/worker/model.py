"""LlamaModel — the data-plane worker: weights, rope tables, paged KV pools, and `forward`.

Public surface and call protocol are the reference's (swiftllm/worker/model.py:18-408):

    model = LlamaModel(engine_config); model.load_weights()
    n = model.profile_num_blocks(); model.init_kvcache_and_swap(n)
    tokens = model.forward(input_ids_list, seq_ids_list, decoding_seq_lens_list)
    model.swap_in_seqs(ids) / model.swap_out_seqs(ids) / model.free_seqs_resources(ids)

so examples/offline.py and the Engine drive it unchanged. Inside, the host side is built for MI355X:
  * all per-forward metadata is planned with numpy (batch_plan.py), packed into ONE pinned buffer and
    shipped with ONE async H2D copy (the reference issues ~10 small synchronous `torch.tensor(...,
    device="cuda")` copies, model.py:272-297);
  * block allocation is decided on a host mirror (block_manager.py): no device sync in `forward`
    except the final D2H of the sampled tokens;
  * rope rows are looked up inside the rotary kernel (no cos/sin gather launches);
  * pure-decode forwards can be captured into hipGraphs, one per batch size, and replayed
    (`EngineConfig.use_hip_graph`): ~400 launches per step collapse into one graph launch;
  * KV pools, block tables and scheduler defaults are sized from the 288 GB of one MI355X.
The compute backend is libswiftllm_hip.so only; without a HIP device this class raises.
"""
import math
import os
import time
from typing import List

import torch

from swiftllm_amd import _hip
from swiftllm_amd.engine_config import EngineConfig
from swiftllm_amd.model_config import LlamaModelConfig
from swiftllm_amd.utils import GB

from .batch_plan import BatchPlan, plan_batch
from .block_manager import BlockManager
from . import decode_engine as _decode_engine
from .infer_state import LlamaInferState
from .kernels.block_swapping import swap_blocks
from .kernels.linear import RawResidual
from .kernels.rmsnorm import fused_add_rmsnorm_from_splitk, rmsnorm_inplace
from .layers.pre_layer import LlamaPreLayer
from .layers.transformer_layer import LlamaTransformerLayer
from .layers.post_layer import LlamaPostLayer
from .weight import load_weights, pack_decode_weights

_DTYPES = {"float16": torch.float16, "fp16": torch.float16, "half": torch.float16,
           "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


def _require_hip_device():
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError(
            "no HIP device visible: swiftllm_amd's data plane runs only on an AMD GPU (gfx950) "
            "through libswiftllm_hip.so; there is no CPU path")
    _hip.load()


class _DecodeGraph:
    """A captured pure-decode forward for one batch size."""
    __slots__ = ("graph", "out_tokens", "logits", "seq_block_size", "num_seq_blocks", "engine")


class _DecodeLookahead:
    """Everything the NEXT decode step of the same batch needs, prepared while the GPU was still busy with this one."""
    __slots__ = ("seq_ids", "lens", "tokens", "plan", "dev")


class LlamaModel:
    """A LLaMA model resident on one GPU, driven by the control plane (or directly, offline)."""

    @torch.inference_mode()
    def __init__(self, engine_config: EngineConfig):
        self.engine_config = engine_config
        self.model_config = LlamaModelConfig.load_from_model_path(engine_config.model_path)
        self.dtype = _DTYPES[getattr(engine_config, "dtype", "float16")]
        self.device = torch.device("cuda")

        self.weight = None
        self._cos_cached = self._sin_cached = None
        self.pre_layer = None
        self.transformer_layers = None
        self.post_layer = None

        self.num_blocks = None
        self.k_cache = self.v_cache = None
        self.k_swap = self.v_swap = None
        self.cpu_block_manager = self.gpu_block_manager = None

        self._meta_host = None      # pinned int32 staging for per-forward metadata
        self._meta_host_np = None
        self._meta_dev = None       # its device twin (fixed address: hipGraph replays read it)
        self._meta_done = None
        self._num_slots = 256        # CUs: one 8-wave paged-attention workgroup each
        self._decode_graphs = {}     # (batch, split width, split count, logits kept) -> _DecodeGraph, in LRU order
        self._lookahead = None       # _DecodeLookahead of the step that is expected next (graph replay only)
        self._decode_lookahead = True    # (tests switch it off to hold the fast path to the plain one)
        # optional callable, invoked by forward() in the calling thread once the step's kernels are enqueued and before
        # the host blocks on the sampled tokens (server/engine.py hides its per-request fan-out behind the running step)
        self.after_launch_hook = None
        self._la_host = self._la_host_np = self._la_done = None
        self._eager_uses_graph_buckets = False   # tests: eager launches at the replay path's split geometry
        # SWL_HOST_PROFILE=1: wall-clock split of forward() on the host (plan / blocks / upload / launch / wait for the
        # tokens), summed over calls — tools read it with host_profile()
        self._host_prof = {} if os.environ.get("SWL_HOST_PROFILE") else None
        self._graph_pool = None
        self._scratch = None
        # the one-sequence persistent decode step (worker/decode_engine.py); None: shape / device / memory / switch say no
        self._engine = None
        self.engine_fallbacks = 0    # steps re-run on the multi-launch path after the engine reported a timed-out hand-off
        self.graph_captures = 0      # hipGraph captures so far (the first at a batch bucket = one warm-up forward + one capture)
        self.graph_capture_s = 0.0   # host seconds spent in them: serving reports both
        self._warmed_buckets = set()

    # ------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def load_weights(self):
        """Load (or synthesise) the weights, build the rope tables and the layer objects."""
        _require_hip_device()
        self.weight = load_weights(self.model_config, self.dtype, self.engine_config.model_path,
                                   self.engine_config.use_dummy, device=self.device,
                                   fuse_qkv=getattr(self.engine_config, "fuse_qkv", False))
        self.repack_decode_weights()
        self._tune_routes()
        self._init_to_get_rotary()
        self._num_slots = torch.cuda.get_device_properties(self.device).multi_processor_count
        side_stream = torch.cuda.Stream()
        self.pre_layer = LlamaPreLayer(self.model_config, self.weight)
        self.transformer_layers = [
            LlamaTransformerLayer(self.model_config, self.engine_config, self.weight.layers[i],
                                  side_stream, i)
            for i in range(self.model_config.num_layers)
        ]
        self.post_layer = LlamaPostLayer(self.model_config, self.weight)
        self.post_layer.skinny = bool(getattr(self.engine_config, "use_skinny_gemm", False))

    def _tune_routes(self):
        """Which side serves the 65..256-token decode projections of THIS model on THIS device is decided here, once, before
        any request (kernels/route_tune.py): the first replica of a node measures the classes the r04 table does not cover
        and writes the answers next to the checkpoint, every other replica reads them."""
        ecfg = self.engine_config
        if not (getattr(ecfg, "use_skinny_gemm", False) and getattr(ecfg, "pack_decode_weights", False)
                and ecfg.max_batch_size > 64):
            return
        from .kernels.linear import tune_wide_routes
        stats = tune_wide_routes(self.weight.projection_tensors(), [lw.up_gate_proj for lw in self.weight.layers[:1]],
                                 self.dtype, self.device, ecfg.model_path)
        if stats.get("measured") or stats.get("read"):
            print(f"[Model] decode-projection routing: {stats['measured']} (shape, bucket) classes measured, "
                  f"{stats['read']} read from {stats['path']}", flush=True)

    @torch.inference_mode()
    def repack_decode_weights(self):
        """(Re)build the MFMA-fragment-order copies the decode GEMMs stream (EngineConfig.pack_decode_weights). Called
        by load_weights; call again after modifying weights in place."""
        ecfg = self.engine_config
        if getattr(ecfg, "pack_decode_weights", False) and getattr(ecfg, "use_skinny_gemm", False):
            # The packed copies double the projection weights' footprint. If that would leave less than a quarter
            # of the usable HBM for activations + KV pool, keep only the row-major weights and say so (a 70B model
            # is 140 GB: it fits the 288 GB part once, not twice).
            proj_bytes = sum(t.numel() * t.element_size() for t in self.weight.projection_tensors())
            free_b, total_b = torch.cuda.mem_get_info(self.device)
            usable = total_b * float(getattr(ecfg, "gpu_mem_utilization", 0.97)) - (total_b - free_b)
            if proj_bytes > 0.75 * usable:
                print(f"[Model] pack_decode_weights disabled: a second {proj_bytes / 2**30:.1f} GiB copy of the "
                      f"projection weights would leave {max(0.0, usable - proj_bytes) / 2**30:.1f} GiB for the KV "
                      f"pool; decode GEMMs stream the row-major weights instead", flush=True)
                ecfg.pack_decode_weights = False
                return
            pack_decode_weights(self.weight)
        self._build_decode_engine()

    def _build_decode_engine(self):
        """(Re)build the persistent one-sequence decode step's weight stream (EngineConfig tuning `decode_engine`): a
        third copy of the layer weights, laid out per CU in the order the kernel consumes it. Skipped — the multi-launch
        path serves batch 1 then — when the model shape or the device is not what csrc/decode_engine.hip is laid out for,
        or when the copy would leave less than a quarter of the usable HBM."""
        ecfg, cfg = self.engine_config, self.model_config
        self._engine = None
        if not (getattr(ecfg, "decode_engine", False) and getattr(ecfg, "use_skinny_gemm", False)):
            return
        num_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if cfg.head_dim != 128 or not _decode_engine.supported(cfg, num_cus):
            return
        need = _decode_engine.stream_bytes(cfg, torch.empty((), dtype=self.dtype).element_size())
        free_b, total_b = torch.cuda.mem_get_info(self.device)
        usable = total_b * float(getattr(ecfg, "gpu_mem_utilization", 0.97)) - (total_b - free_b)
        if need > 0.75 * usable:
            print(f"[Model] decode_engine disabled: its {need / 2**30:.1f} GiB weight stream would leave "
                  f"{max(0.0, usable - need) / 2**30:.1f} GiB for the KV pool", flush=True)
            return
        self._engine = _decode_engine.DecodeEngine(cfg, self.weight, self.dtype, self.device)

    @torch.inference_mode()
    def profile_num_blocks(self) -> int:
        """How many KV blocks fit: run a forged worst-case prefill without touching the KV cache,
        read the device's high-water mark, give the rest (up to gpu_mem_utilization) to the pool.
        Reference: model.py:94-131."""
        ecfg = self.engine_config
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        num_tokens, batch = ecfg.max_tokens_in_batch, ecfg.max_batch_size
        lens = [num_tokens // batch] * batch
        lens[-1] += num_tokens % batch
        self.k_cache = self.v_cache = None
        self.forward([[0] * n for n in lens], list(range(batch)), [], ignore_kvcache=True)
        torch.cuda.synchronize()
        free_memory, total_memory = torch.cuda.mem_get_info()
        peak_memory = total_memory - free_memory
        usable = total_memory * ecfg.gpu_mem_utilization
        print(f"[Model.profile] GPU total memory: {total_memory / GB:.2f} GB, "
              f"runtime peak memory: {peak_memory / GB:.2f} GB")
        if usable < peak_memory:
            raise RuntimeError(
                f"Peak memory {peak_memory / GB:.2f} GB exceeds usable memory {usable / GB:.2f} GB "
                f"({total_memory / GB:.2f} GB * {ecfg.gpu_mem_utilization})")
        block_bytes = ecfg.block_size * self.model_config.get_kvslot_size(self.dtype)
        # the two block tables (GPU + CPU manager) are allocated after this point: budget them
        table_bytes = 2 * ecfg.max_seqs_in_block_table * ecfg.max_blocks_per_seq * 4
        # ... and so is the private memory pool of the decode hipGraphs (captured on first use, AFTER the KV pool took
        # its share): a max_batch_size decode forward's activations, [B, vocab] logits (+ fp32 argmax candidates) and
        # split-K scratch. The eager prefill peak above does not contain it.
        graph_bytes = 0
        if getattr(ecfg, "use_hip_graph", False):
            graph_bytes = self._measure_decode_graph_bytes()
            free_memory, _ = torch.cuda.mem_get_info()      # (the measurement may have grown the regular pool as well)
            peak_memory = max(peak_memory, total_memory - free_memory)
        num_blocks = math.floor((usable - peak_memory - table_bytes - graph_bytes) / block_bytes)
        torch.cuda.empty_cache()
        return max(num_blocks, 0)

    def _measure_decode_graph_bytes(self) -> int:
        """What the private memory pool of the decode hipGraphs will hold: MEASURED (r03-r04 used a formula) as
        the allocator's high-water mark over one eager pure-decode forward at max_batch_size on a throw-away KV pool of one
        block per sequence — a capture allocates what the eager forward allocates, into its own pool, and all captures share
        ONE pool (_forward_decode_graph), so the largest batch sets its size. 25 % + 16 MiB on top for the allocator's
        rounding. Falls back to the r04 formula when the throw-away pool cannot be built."""
        cfg, ecfg = self.model_config, self.engine_config
        # the largest CAPTURED batch: max_batch_size rounded up to its replay bucket (up to 15 inert rows more)
        b = min(self._decode_batch_bucket(int(ecfg.max_batch_size)), int(ecfg.max_seqs_in_block_table))
        formula = b * cfg.vocab_size * 8 + b * (cfg.hidden_size + cfg.ffn_inter_dim) * 64 + (64 << 20)
        if b <= 0:
            return formula
        saved = (self.k_cache, self.v_cache, self.gpu_block_manager, self.cpu_block_manager, ecfg.use_hip_graph)
        try:
            shape = (b, cfg.num_layers, cfg.num_kv_heads, ecfg.block_size, cfg.head_dim)
            self.k_cache = torch.zeros(shape, dtype=self.dtype, device=self.device)
            self.v_cache = torch.zeros(shape, dtype=self.dtype, device=self.device)
            self.gpu_block_manager = BlockManager("GPU", b, ecfg.max_seqs_in_block_table, ecfg.max_blocks_per_seq,
                                                  ecfg.block_size, self.device)
            ecfg.use_hip_graph = False
            self.forward([[0]] * b, list(range(b)), [1] * b)        # (also sizes the persistent split-K workspace)
            torch.cuda.synchronize()
            self.gpu_block_manager.free_blocks_for_seqs(list(range(b)))
            base = torch.cuda.memory_allocated()
            torch.cuda.reset_peak_memory_stats()
            self.forward([[0]] * b, list(range(b)), [1] * b)
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() - base
            # + what every cached graph keeps alive in the shared pool after its capture: its sampled-token tensor (the
            # [B, vocab] logits are retained only while a test taps them — _forward_decode_graph), rounded to the
            # allocator's 512-byte granule, for a full replay cache
            retained = self._MAX_DECODE_GRAPHS * (-(-b * 8 // 512) * 512 + 512)
            measured = int(peak * 1.25) + (16 << 20) + retained
            print(f"[Model.profile] decode activations at batch {b}: {peak / 2**20:.1f} MiB "
                  f"(hipGraph pool reserve {measured / 2**20:.1f} MiB; r04 formula: {formula / 2**20:.1f} MiB)")
            return measured
        except Exception as exc:     # noqa: BLE001 — sizing aid only: any failure falls back to the formula
            print(f"[Model.profile] decode-pool measurement failed ({type(exc).__name__}: {exc}); using the formula")
            return formula
        finally:
            self.k_cache, self.v_cache, self.gpu_block_manager, self.cpu_block_manager, ecfg.use_hip_graph = saved
            self._lookahead = None
            self._drop_decode_graphs()
            torch.cuda.empty_cache()

    @torch.inference_mode()
    def init_kvcache_and_swap(self, num_blocks: int):
        """Allocate the GPU KV pools, the host swap pools and both block managers.
        Reference: model.py:134-175."""
        _require_hip_device()
        cfg, ecfg = self.model_config, self.engine_config
        self.num_blocks = num_blocks
        shape = (num_blocks, cfg.num_layers, cfg.num_kv_heads, ecfg.block_size, cfg.head_dim)
        # zeros, not empty: stale NaNs in never-written slots would poison 0 * NaN products
        self.k_cache = torch.zeros(shape, dtype=self.dtype, device=self.device)
        self.v_cache = torch.zeros(shape, dtype=self.dtype, device=self.device)
        swap_shape = (ecfg.num_cpu_blocks,) + shape[1:]
        pin = getattr(ecfg, "pin_swap_memory", True) and ecfg.num_cpu_blocks > 0
        try:
            self.k_swap = torch.zeros(swap_shape, dtype=self.dtype, device="cpu", pin_memory=pin)
            self.v_swap = torch.zeros(swap_shape, dtype=self.dtype, device="cpu", pin_memory=pin)
        except RuntimeError:    # host cannot pin that much: pageable pools, as the reference has
            self.k_swap = torch.zeros(swap_shape, dtype=self.dtype, device="cpu")
            self.v_swap = torch.zeros(swap_shape, dtype=self.dtype, device="cpu")
        self.gpu_block_manager = BlockManager("GPU", num_blocks, ecfg.max_seqs_in_block_table,
                                              ecfg.max_blocks_per_seq, ecfg.block_size, self.device)
        self.cpu_block_manager = BlockManager("CPU", ecfg.num_cpu_blocks,
                                              ecfg.max_seqs_in_block_table, ecfg.max_blocks_per_seq,
                                              ecfg.block_size, self.device)
        self._drop_decode_graphs()
        self._lookahead = None

    def _init_to_get_rotary(self):
        """cos/sin tables [positions, head_dim/2] in the model dtype. Formulas (including the
        non-HuggingFace frequency split used for dict-valued rope_scaling) follow the reference,
        model.py:177-225, evaluated in fp32 on the device."""
        cfg = self.model_config
        dev, f32 = self.device, torch.float32
        base, dim = cfg.rope_theta, cfg.head_dim
        scaling = cfg.rope_scaling
        if isinstance(scaling, dict):
            factor = scaling.get("factor", 4.0)
            low = scaling.get("low_freq_factor", 1.0)
            high = scaling.get("high_freq_factor", 1.0)
            orig = scaling.get("original_max_position_embeddings", cfg.max_position_embeddings)
            t = torch.arange(int(orig * factor) + 128, device=dev, dtype=f32)
            split = int((dim // 2) * low / (low + high))
            inv_low = 1.0 / (base ** (torch.arange(0, split * 2, 2, device=dev, dtype=f32) / dim))
            inv_high = 1.0 / (base ** (torch.arange(split * 2, dim, 2, device=dev, dtype=f32) / dim))
            freqs = torch.cat([torch.outer(t / low, inv_low), torch.outer(t / high, inv_high)], dim=-1)
        else:
            inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, device=dev, dtype=f32) / dim))
            t = torch.arange(cfg.max_position_embeddings * scaling + 128, device=dev, dtype=f32) / scaling
            freqs = torch.outer(t, inv_freq)
        self._cos_cached = torch.cos(freqs).to(self.dtype).contiguous()
        self._sin_cached = torch.sin(freqs).to(self.dtype).contiguous()

    # ------------------------------------------------------------------------------------------------
    def _upload_plan(self, plan: BatchPlan):
        """Pack the plan's int32 arrays into pinned memory, one async H2D copy, return device views."""
        layout, total = plan.packed_layout()
        if self._meta_done is not None:
            self._meta_done.synchronize()
        if self._meta_host is None or self._meta_host.numel() < total:
            ecfg = self.engine_config
            cap = max(total, 2 * ecfg.max_tokens_in_batch + 8 * ecfg.max_batch_size + 64)
            self._meta_host = torch.empty(cap, dtype=torch.int32, pin_memory=True)
            self._meta_host_np = self._meta_host.numpy()
            self._meta_dev = torch.empty(cap, dtype=torch.int32, device=self.device)
            self._meta_done = torch.cuda.Event()
            self._drop_decode_graphs()     # captured graphs point into the old buffer
        plan.pack_into(self._meta_host_np)
        self._meta_dev[:total].copy_(self._meta_host[:total], non_blocking=True)
        self._meta_done.record()
        return {name: self._meta_dev[off:off + n] for name, off, n in layout}

    def _make_infer_state(self, plan: BatchPlan, dev: dict, ignore_kvcache: bool) -> LlamaInferState:
        nsb = plan.num_seq_blocks
        if nsb > 1:
            need = _hip.scratch_bytes(plan.num_decoding_seqs, self.model_config.num_q_heads,
                                      self.model_config.head_dim, nsb) // 4
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float32, device=self.device)
                self._drop_decode_graphs()
        return LlamaInferState(
            batch_size=plan.batch_size, num_tokens=plan.num_tokens,
            seq_ids=dev["seq_ids"], softmax_scale=self.model_config.head_dim ** -0.5,
            num_prefill_seqs=plan.num_prefill_seqs, num_prefill_tokens=plan.num_prefill_tokens,
            prefill_seq_start_locs=dev["prefill_start_locs_with_end"][:plan.num_prefill_seqs],
            prefill_seq_start_locs_with_end=dev["prefill_start_locs_with_end"],
            prefill_seq_lens=dev["prefill_seq_lens"], max_prefill_len=plan.max_prefill_len,
            num_decoding_seqs=plan.num_decoding_seqs, decoding_seq_lens=dev["decoding_seq_lens"],
            max_decoding_len=plan.max_decoding_len,
            seq_block_size=plan.seq_block_size, num_seq_blocks=nsb,
            position_cos=self._cos_cached, position_sin=self._sin_cached,
            ignore_kvcache=ignore_kvcache,
            position_indices=dev["position_indices"], last_token_indices=dev["last_token_indices"],
            paged_attn_scratch=self._scratch if nsb > 1 else None)

    @torch.inference_mode()
    def _forward(self, input_ids: torch.Tensor, infer_state: LlamaInferState) -> torch.Tensor:
        """Embedding -> L transformer blocks -> final norm / lm_head / argmax.
        Reference: model.py:228-249."""
        if self._engine_applies(infer_state):
            # ONE sequence, pure decode: embedding + all transformer blocks in one persistent launch
            x = self._engine.step(self.k_cache, self.v_cache, self.gpu_block_manager.block_table, input_ids,
                                  infer_state.seq_ids, infer_state.decoding_seq_lens, self._cos_cached, self._sin_cached,
                                  self.engine_config.max_blocks_per_seq)
            rmsnorm_inplace(x, self.weight.final_norm, self.model_config.rms_norm_eps)
            return self.post_layer.forward_normed(x, out=self._engine.tok_err[:1])
        x = self.pre_layer.forward(input_ids)
        residual = torch.zeros_like(x)
        block_table = None if infer_state.ignore_kvcache else self.gpu_block_manager.block_table
        for layer in self.transformer_layers:
            x = layer.forward(x, residual, self.k_cache, self.v_cache, block_table, infer_state)
        if isinstance(x, RawResidual):          # the last down projection added itself into the residual (rows_decode):
            x = residual.clone()                # pure decode, every row a last token — final norm of the residual rows
            rmsnorm_inplace(x, self.weight.final_norm, self.model_config.rms_norm_eps)
            return self.post_layer.forward_normed(x)
        if not isinstance(x, torch.Tensor):     # the last down projection left as split-K partials
            if infer_state.num_prefill_seqs == 0:
                # pure decode: every row is a last token — reduce + residual add + final norm in one launch
                x = fused_add_rmsnorm_from_splitk(x, residual, self.weight.final_norm, self.model_config.rms_norm_eps)
                return self.post_layer.forward_normed(x)
            x = x.materialize()
        x += residual
        return self.post_layer.forward(x, infer_state)

    def _engine_applies(self, st) -> bool:
        return (self._engine is not None and st.num_prefill_seqs == 0 and st.batch_size == 1
                and st.num_decoding_seqs == 1 and not st.ignore_kvcache and self.k_cache is not None)

    def _engine_failed(self, code: int):
        """A hand-off of the persistent step timed out (bounded spins: the launch ended, its results are garbage). Say so,
        clear the workspace, and leave batch 1 to the multi-launch HIP path from here on."""
        print(f"[Model] decode engine reported error {code & 0xff} (CU {code >> 8}): falling back to the multi-launch "
              f"decode path for one-sequence steps", flush=True)
        torch.cuda.synchronize()
        self._engine.reset()
        self._engine = None
        self._drop_decode_graphs()
        self._lookahead = None
        self.engine_fallbacks += 1

    # ---- hipGraph replay of pure-decode steps ------------------------------------------------------------
    def _drop_decode_graphs(self):
        """Forget every captured decode graph (their buffers moved, or the pool / engine they were captured against is gone)
        AND the private memory pool they shared. The pool handle must go with them: once its last graph is destroyed the
        caching allocator marks the pool freeable (use_count 0) but keeps it on its books until the next cache flush, and a
        new capture into that id trips `it->second->use_count > 0` (HIPCachingAllocator.cpp) — met in r06 by the serving
        sweep (tools/serve_bench.py --sweep: the flash-decoding scratch grew after graphs had been captured) and by
        bench.py's engine leg. The next capture opens a fresh pool; the old one's memory returns to the device on the
        allocator's next flush."""
        self._decode_graphs.clear()
        self._graph_pool = None
        self._warmed_buckets.clear()     # (workspaces may have been re-sized: warm up again before the next capture)

    # LRU bound of the replay cache. Since r06b a cached graph pins only its sampled-token tensor in the shared pool (the
    # [B, vocab] logits are kept only under a test's tap), so the bound is about host-side graph objects, not HBM. 48 (r05)
    # thrashed under a serving sweep with ShareGPT-like lengths at max batch 256: 14 batch buckets x ~5 single-split widths
    # + the split geometries of small batches = well over 100 live keys, 23-53 captures per 1000 forwards at 40-200 req/s
    # (profiles/r06b_serve_sweep_sharegpt_*.jsonl).
    _MAX_DECODE_GRAPHS = 256

    def _graph_bucket(self, plan: BatchPlan):
        """Captured launch geometry must cover every replay, and the number of distinct geometries a long-running
        server meets must stay small (every new one costs a warm-up run + a capture, and memory): the split width
        — `select_seq_block_size` drifts in 64-token steps as sequences grow — is rounded up to a multiple of
        1/16..1/32 of itself (at most 16 widths per octave), and the split count is taken for the longest sequence
        rounded up to its next 64-token boundary and then to a power of two (surplus workgroups exit on their first
        instruction). A single split stays a single split (its width only has to cover the longest sequence).
        (r01/r02 rounded the width to a POWER of two: at Llama-2-7B dims, 4 x 16.4k tokens, that turned the balanced
        8256 + 8139 split into 16384 + 11 — half the workgroups idle, the attention kernel 224 instead of 175 us and
        graph replay 12 % slower than eager launches.)"""
        sbs = int(plan.seq_block_size)
        if plan.num_seq_blocks > 1:
            quantum = max(64, 1 << max(0, sbs.bit_length() - 5))
            sbs = -(-sbs // quantum) * quantum
        horizon = -(-(plan.max_decoding_len + 1) // 64) * 64
        nsb_cap = -(-horizon // sbs)
        if nsb_cap > 1:
            nsb_cap = 1 << (nsb_cap - 1).bit_length()
        else:
            # one split: any width >= the longest sequence is the same launch; key on a power of two of it
            sbs = max(256, 1 << (horizon - 1).bit_length())
        return sbs, nsb_cap

    def _decode_batch_bucket(self, batch: int) -> int:
        """Captured batch size that serves a pure-decode batch of `batch` sequences. The replay cache is keyed on the batch
        size, and a server's batch drifts by one sequence at a time: keyed on the EXACT size (r01-r04) every new size paid a
        warm-up forward plus a capture. Sizes are rounded up — to a multiple of 8 up to 32 sequences, of 16 up to 64, of 32
        beyond (the projections' cost moves in 32-token blocks there) — and the surplus rows are INERT: length-0 sequences, which every
        decode kernel treats as a no-op (paged attention: the workgroup exits before its prologue; rotary / KV store: nothing
        read, nothing stored; the projections, norms and the sampler are row-independent, so whatever an inert row holds
        never reaches a real one). 1 and 2 stay exact (their own launch path). The reference has no graphs
        (swiftllm/worker/model.py:228-249 launches eagerly); results are those of the exact-size launch."""
        if batch <= 2:
            return batch
        # r06b: 16 between 32 and 64 (the medium kernels serve two token blocks whatever the count), 32 beyond 64 (the wide
        # GEMMs and the library work in 32-token blocks: 129 -> 160 costs what 144 did) — 14 buckets up to 256 instead of 22
        step = 8 if batch <= 32 else 16 if batch <= 64 else 32
        return -(-batch // step) * step

    def _plan_decode(self, seq_ids_list: List[int], lens: List[int], tokens, pad: bool) -> BatchPlan:
        """Plan of a pure-decode step; `pad`: rounded up to its batch bucket with inert rows (token 0, length 0)."""
        b = len(lens)
        bp = self._decode_batch_bucket(b) if pad else b
        ids = tokens if tokens is not None else [(0,)] * b
        if bp > b:
            ids = list(ids) + [(0,)] * (bp - b)
            seq_ids_list = list(seq_ids_list) + [seq_ids_list[0]] * (bp - b)
            lens = list(lens) + [0] * (bp - b)
        plan = plan_batch(ids, seq_ids_list, lens, self.model_config.num_kv_heads, self._num_slots)
        plan.num_real_seqs = b
        return plan

    def _forward_decode_graph(self, plan: BatchPlan, dev: dict) -> torch.Tensor:
        sbs, nsb_cap = self._graph_bucket(plan)
        use_engine = self._engine is not None and plan.batch_size == 1
        tap = self.post_layer.logits_tap
        tap_len = len(tap) if tap is not None else 0
        # (the engine has one launch geometry; a graph captured while a test taps the logits keeps its [B, vocab] tensor
        # alive in the shared pool, one captured without the tap does not — up to 65 MB per graph at 256 sequences)
        key = ((1, 0, 0) if use_engine else (plan.batch_size, sbs, nsb_cap)) + (tap is not None,)
        entry = self._decode_graphs.pop(key, None)
        plan.seq_block_size, plan.num_seq_blocks = sbs, nsb_cap
        if entry is None:
            self.graph_captures += 1
            t_cap = time.perf_counter()
            state = self._make_infer_state(plan, dev, False)
            # one eager run on a side stream first (library handles, workspaces, allocator pools): nothing may be lazily
            # initialised while the stream is capturing. Once per BATCH BUCKET (r06b): a new split geometry at a batch size
            # that has been captured before launches the same kernels on the same shapes — only the attention grid moves.
            warm_key = (key[0], use_engine, tap is not None)
            if warm_key not in self._warmed_buckets:
                warm = torch.cuda.Stream()
                warm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(warm):
                    self._forward(dev["input_ids"], state)
                torch.cuda.current_stream().wait_stream(warm)
                self._warmed_buckets.add(warm_key)
            while len(self._decode_graphs) >= self._MAX_DECODE_GRAPHS:      # least recently used first
                self._decode_graphs.pop(next(iter(self._decode_graphs)))
            if self._graph_pool is None:
                # ONE private memory pool for all captures: graphs never run concurrently (one stream), so they can
                # share their activation memory instead of each pinning its own after the KV pool took the rest
                self._graph_pool = torch.cuda.graph_pool_handle()
            entry = _DecodeGraph()
            entry.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(entry.graph, pool=self._graph_pool):
                entry.out_tokens = self._forward(dev["input_ids"], state)
            entry.logits = self.post_layer.last_logits if tap is not None else None
            if tap is None:
                self.post_layer.last_logits = None      # (drop the last reference: the buffer returns to the shared pool)
            entry.seq_block_size, entry.num_seq_blocks = sbs, nsb_cap
            entry.engine = use_engine
            if tap is not None:
                del tap[tap_len:]               # what the warm-up run and the capture appended
            self.graph_capture_s += time.perf_counter() - t_cap
        self._decode_graphs[key] = entry        # (re)inserted last = most recently used
        entry.graph.replay()
        n = plan.real_seqs
        if tap is not None:                     # (tests) one entry per forward, as on the eager path; a copy: the
            tap.append(entry.logits[:n].clone())    # graph's own tensor is overwritten by the next replay
        return entry.out_tokens[:n]

    # ---- decode look-ahead: the host side of step k+1 runs while the GPU executes step k ---------------------------------
    def _prepare_next_decode(self, plan: BatchPlan, seq_ids_list: List[int], tokens_dev: torch.Tensor):
        """Called right after the graph of a pure-decode step was launched, before the host blocks on its tokens. A
        generation loop calls forward again with the same sequences, every length + 1 and the tokens just sampled: plan
        that step now (numpy) and enqueue — behind the running graph, on the same stream — the H2D copy of its metadata and
        a device copy of the sampled tokens into its input-id slots. If the next call is that step (forward checks), it
        only takes its KV blocks and launches: ~40 us of host work per step the GPU no longer waits for
        (tools/host_overhead.py: plan 18 + upload 24 us at batch 32). If it is not, nothing is lost: the ordinary path
        re-plans and re-uploads. (Blocks are NOT taken early: a scheduler that counts the pool down to its last block must
        find the allocator exactly where its own books say it is.)"""
        b = plan.real_seqs
        next_lens = [n + 1 for n in plan.seq_lengths_list[:b]]
        if max(next_lens) > self._cos_cached.shape[0]:
            return None
        nxt = self._plan_decode(seq_ids_list, next_lens, None, True)
        layout, total = nxt.packed_layout()
        if self._la_host is None or self._la_host.numel() < total:
            self._la_host = torch.empty(max(total, self._meta_host.numel()), dtype=torch.int32, pin_memory=True)
            self._la_host_np = self._la_host.numpy()
            self._la_done = torch.cuda.Event()
        else:
            self._la_done.synchronize()     # (the previous look-ahead copy out of this pinned buffer: long done)
        nxt.pack_into(self._la_host_np)
        self._meta_dev[:total].copy_(self._la_host[:total], non_blocking=True)
        self._la_done.record()
        dev = {name: self._meta_dev[off:off + n] for name, off, n in layout}
        dev["input_ids"][:b].copy_(tokens_dev)  # int64 -> int32 on the device: the host has not seen them yet (inert rows: 0)
        la = _DecodeLookahead()
        la.seq_ids, la.lens, la.plan, la.dev, la.tokens = list(seq_ids_list), next_lens, nxt, dev, None
        return la

    def _take_lookahead(self, input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache):
        """The prepared step, if this call is it (any other call drops the preparation)."""
        la, self._lookahead = self._lookahead, None
        if (la is None or la.tokens is None or ignore_kvcache or not self._decode_lookahead
                or not getattr(self.engine_config, "use_hip_graph", False)
                or len(input_ids_list) != len(la.lens) or list(decoding_seq_lens_list) != la.lens
                or list(seq_ids_list) != la.seq_ids):
            return None
        for ids, tok in zip(input_ids_list, la.tokens):
            if len(ids) != 1 or ids[0] != tok:
                return None
        return la

    # ------------------------------------------------------------------------------------------------
    def _tokens_to_host(self, tokens: torch.Tensor) -> List[int]:
        """The step's one D2H copy (and its one sync). A step of the persistent decode engine brings its error word back in
        the same copy (DecodeEngine.tok_err = [token, error])."""
        eng = self._engine
        if eng is not None and tokens.data_ptr() == eng.tok_err.data_ptr():
            vals = eng.tok_err.tolist()
            if vals[1] != 0:
                raise _decode_engine.DecodeEngineError(str(vals[1]))
            return vals[:1]
        return tokens.tolist()

    @torch.inference_mode()
    def forward(self, input_ids_list: List[List[int]], seq_ids_list: List[int],
                decoding_seq_lens_list: List[int], ignore_kvcache: bool = False) -> List[int]:
        """One iteration: prefill sequences first, then decoding sequences (one token each, their
        lengths in `decoding_seq_lens_list` INCLUDE that token). Returns the greedy next token of
        every sequence. Reference: model.py:252-359."""
        if len(input_ids_list) == 0:
            return []   # the reference's idle engine calls forward([], [], []) in a loop
        _require_hip_device()
        try:
            return self._forward_step(input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache)
        except _decode_engine.DecodeEngineError as exc:
            # bounded hand-off timed out inside the persistent one-sequence step: its KV writes are re-done below with the
            # same values by the multi-launch HIP path, which serves one-sequence steps from here on
            self._engine_failed(int(str(exc)))
            return self._forward_step(input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache)

    def _forward_step(self, input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache):
        prof = self._host_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        la = self._take_lookahead(input_ids_list, seq_ids_list, decoding_seq_lens_list, ignore_kvcache)
        if la is not None:      # the step that was prepared while the previous one ran: blocks, then the launch
            self.gpu_block_manager.allocate_blocks_for_seqs(la.seq_ids, la.lens)   # (the real sequences only)
            tokens = self._forward_decode_graph(la.plan, la.dev)
            nxt = self._prepare_next_decode(la.plan, la.seq_ids, tokens)
            if self.after_launch_hook is not None:
                self.after_launch_hook()
            t4 = time.perf_counter() if prof is not None else 0.0
            out = self._tokens_to_host(tokens)
            if nxt is not None:
                nxt.tokens = out
            self._lookahead = nxt
            if prof is not None:
                for name, dt in (("launch", t4 - t0), ("wait_tokens", time.perf_counter() - t4), ("calls", 1.0),
                                 ("lookahead_hits", 1.0)):
                    prof[name] = prof.get(name, 0.0) + dt
            return out
        num_real = len(input_ids_list)
        graph_step = (len(decoding_seq_lens_list) == num_real and not ignore_kvcache
                      and getattr(self.engine_config, "use_hip_graph", False))
        if graph_step or (self._eager_uses_graph_buckets and len(decoding_seq_lens_list) == num_real and not ignore_kvcache):
            # pure decode on the replay path: the batch is rounded up to its bucket with inert rows (_decode_batch_bucket)
            plan = self._plan_decode(seq_ids_list, list(decoding_seq_lens_list), input_ids_list, True)
        else:
            plan = plan_batch(input_ids_list, seq_ids_list, decoding_seq_lens_list,
                              self.model_config.num_kv_heads, self._num_slots)
            plan.num_real_seqs = num_real
        longest = max(plan.seq_lengths_list)
        if longest > self._cos_cached.shape[0]:
            # the reference indexes its rope cache out of range here (model.py:350); fail on the host
            raise RuntimeError(
                f"sequence length {longest} exceeds the rotary table ({self._cos_cached.shape[0]} positions = "
                "max_position_embeddings * rope_scaling + 128); use a checkpoint with rope scaling")
        t1 = time.perf_counter() if prof is not None else 0.0
        if not ignore_kvcache:
            self.gpu_block_manager.allocate_blocks_for_seqs(seq_ids_list, plan.seq_lengths_list[:num_real])
        t2 = time.perf_counter() if prof is not None else 0.0
        dev = self._upload_plan(plan)
        t3 = time.perf_counter() if prof is not None else 0.0
        pure_decode = plan.num_prefill_seqs == 0
        nxt = None
        if (pure_decode and not ignore_kvcache and getattr(self.engine_config, "use_hip_graph", False)):
            tokens = self._forward_decode_graph(plan, dev)
            if self._decode_lookahead:
                nxt = self._prepare_next_decode(plan, list(seq_ids_list), tokens)
        else:
            if pure_decode and not ignore_kvcache and self._eager_uses_graph_buckets:
                plan.seq_block_size, plan.num_seq_blocks = self._graph_bucket(plan)
            tokens = self._forward(dev["input_ids"], self._make_infer_state(plan, dev, ignore_kvcache))[:num_real]
            tap = self.post_layer.logits_tap
            if tap and plan.batch_size > num_real:      # (tests) a padded eager step: the real rows only, as on replay
                tap[-1] = tap[-1][:num_real]
        if self.after_launch_hook is not None:
            self.after_launch_hook()
        if prof is None:
            out = self._tokens_to_host(tokens)
            if nxt is not None:
                nxt.tokens = out
            self._lookahead = nxt
            return out
        t4 = time.perf_counter()
        out = self._tokens_to_host(tokens)
        t5 = time.perf_counter()
        if nxt is not None:
            nxt.tokens = out
        self._lookahead = nxt
        for name, dt in (("plan", t1 - t0), ("blocks", t2 - t1), ("upload", t3 - t2), ("launch", t4 - t3),
                         ("wait_tokens", t5 - t4), ("calls", 1.0)):
            prof[name] = prof.get(name, 0.0) + dt
        return out

    def host_profile(self, reset: bool = True) -> dict:
        """Mean host seconds per forward() call by section (SWL_HOST_PROFILE=1), {} when profiling is off."""
        prof = self._host_prof or {}
        n = prof.get("calls", 0.0)
        out = {k: v / n for k, v in prof.items() if k != "calls"} if n else {}
        if reset and self._host_prof is not None:
            self._host_prof.clear()
        return out

    # ---- swapping ----------------------------------------------------------------------------------------
    def _swap(self, seq_ids_list: List[int], is_swap_in: bool):
        """Move every block of the given sequences between the GPU pool and the host swap pool.
        Reference: model.py:361-379. Block ids come from the host mirrors — no device read-back."""
        if len(seq_ids_list) == 0:
            return
        self._lookahead = None
        src_mgr = self.cpu_block_manager if is_swap_in else self.gpu_block_manager
        dst_mgr = self.gpu_block_manager if is_swap_in else self.cpu_block_manager
        counts = src_mgr.get_num_allocated_blocks_host(seq_ids_list)
        src_ids = [b for s in seq_ids_list for b in src_mgr.get_block_ids_host(s)]
        src_mgr.gather_allocated_blocks_and_free(seq_ids_list)
        dst_mgr.allocate_blocks_for_seqs(seq_ids_list, [c * self.engine_config.block_size for c in counts])
        dst_ids = [b for s in seq_ids_list for b in dst_mgr.get_block_ids_host(s)]
        swap_blocks(src_ids, dst_ids, is_swap_in, self.k_cache, self.v_cache, self.k_swap, self.v_swap)

    @torch.inference_mode()
    def swap_in_seqs(self, seq_ids_list: List[int]):
        """Bring the sequences' KV blocks back from the host swap pool."""
        self._swap(seq_ids_list, True)

    @torch.inference_mode()
    def swap_out_seqs(self, seq_ids_list: List[int]):
        """Evict the sequences' KV blocks to the host swap pool."""
        self._swap(seq_ids_list, False)

    @torch.inference_mode()
    def free_seqs_resources(self, seq_ids_list: List[int]):
        """Release everything the sequences hold, on both pools. Reference: model.py:402-408."""
        if len(seq_ids_list) == 0:
            return
        self._lookahead = None
        self.gpu_block_manager.free_blocks_for_seqs(seq_ids_list)
        self.cpu_block_manager.free_blocks_for_seqs(seq_ids_list)
