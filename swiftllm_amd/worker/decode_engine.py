"""The one-sequence decode step as ONE persistent launch — host side of csrc/decode_engine.hip.

What the reference does per layer in ~13 launches (swiftllm/worker/layers/transformer_layer.py:31-130, looped by
swiftllm/worker/model.py:228-249) and the multi-launch path here in 5-6, the engine does for all layers in one: 256
workgroups, each streaming ITS rows of every projection from one contiguous run of HBM. This module owns

  * the packing of that stream (`pack_engine_layer`: pure tensor permutations, testable on CPU —
    tests/test_host_logic.py holds it to the kernel's addressing),
  * the hand-off workspace and the persistent output buffers,
  * the launch (`DecodeEngine.step`) and the error protocol (a timed-out hand-off poisons the workspace; the model then
    drops back to the multi-launch path — another HIP path, never a CPU fallback).
"""
import os

import torch

from swiftllm_amd import _hip

NUM_CUS = 256               # the stream is laid out for the 256 CUs of one MI355X (csrc/decode_engine.hip kCUs)
SLOT_ELEMS = 8192           # one 16 KiB ring slot
ROWS_PER_GROUP = 8
K_CHUNK = 1024


def _pack_rows(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] (N = 256 * R, R % 8 == 0, K % 1024 == 0) -> [256, (R/8) * (K/1024) * 8192]: per CU the slots of its R rows
    in (row group, k-chunk) order; a slot = 16 pieces of [8 rows][8 k-sub-blocks][8 elements] (lane l of a piece = row
    l // 8, k-sub-block l % 8), so a consumer wave's 16-byte LDS read at lane * 16 is its 8 consecutive k."""
    n, k = w.shape
    r = n // NUM_CUS
    assert n % (NUM_CUS * ROWS_PER_GROUP) == 0 and k % K_CHUNK == 0, (n, k)
    g, kj = r // ROWS_PER_GROUP, k // K_CHUNK
    v = w.reshape(NUM_CUS, g, ROWS_PER_GROUP, kj, 16, 8, 8)        # cu, group, row, k-chunk, piece, k-sub, element
    v = v.permute(0, 1, 3, 4, 2, 5, 6)                              # cu, group, k-chunk, piece, row, k-sub, element
    return v.reshape(NUM_CUS, g * kj * SLOT_ELEMS)


def pack_engine_layer(qkv: torch.Tensor, o: torch.Tensor, up_gate: torch.Tensor, down: torch.Tensor) -> torch.Tensor:
    """One layer's stream [256, slots_per_layer * 8192]: qkv | o | up rows, then the matching gate rows | down.
    `up_gate` is the reference's [up ; gate] concatenation (swiftllm/worker/weight.py:133)."""
    inter = up_gate.shape[0] // 2
    return torch.cat((_pack_rows(qkv), _pack_rows(o), _pack_rows(up_gate[:inter]), _pack_rows(up_gate[inter:]),
                      _pack_rows(down)), dim=1).contiguous()


def supported(cfg, num_cus: int) -> bool:
    lib = _hip.load()
    return bool(lib.swl_decode_engine_supported(cfg.hidden_size, cfg.num_q_heads, cfg.num_kv_heads, cfg.head_dim,
                                                cfg.ffn_inter_dim, num_cus))


def stream_bytes(cfg, itemsize: int) -> int:
    spl = _hip.load().swl_decode_engine_slots_per_layer(cfg.hidden_size, cfg.num_q_heads, cfg.num_kv_heads,
                                                        cfg.ffn_inter_dim)
    return cfg.num_layers * NUM_CUS * spl * SLOT_ELEMS * itemsize


class DecodeEngineError(_hip.HipLibraryError):
    """A hand-off inside the persistent decode step timed out (code = wait kind | CU << 8)."""


class DecodeEngine:
    def __init__(self, model_config, weight, dtype: torch.dtype, device: torch.device):
        cfg = self.cfg = model_config
        self.dtype, self.device = dtype, device
        lib = _hip.load()
        self.spl = int(lib.swl_decode_engine_slots_per_layer(cfg.hidden_size, cfg.num_q_heads, cfg.num_kv_heads,
                                                             cfg.ffn_inter_dim))
        assert self.spl > 0
        self.stream = torch.empty((cfg.num_layers, NUM_CUS, self.spl * SLOT_ELEMS), dtype=dtype, device=device)
        self.norms = torch.empty((cfg.num_layers, 2, cfg.hidden_size), dtype=dtype, device=device)
        self.repack(weight)
        self.ws_bytes = int(lib.swl_decode_engine_workspace_bytes(cfg.hidden_size, cfg.num_q_heads, cfg.num_kv_heads,
                                                                  cfg.ffn_inter_dim))
        self.ws = torch.empty(self.ws_bytes // 8, dtype=torch.int64, device=device)
        self.resid = torch.zeros((1, cfg.hidden_size), dtype=dtype, device=device)
        # [0] the sampled token (written by the sampler), [1] the engine's error word: ONE D2H copy brings both back
        self.tok_err = torch.zeros(2, dtype=torch.int64, device=device)
        self.debug_stamps = None
        # bit 0: thin the weight stream while a consumer wave sweeps granules (csrc/decode_engine.hip); SWL_ENGINE_FLAGS
        # overrides it for A/B measurements (tools/engine_trace.py)
        self.flags = int(os.environ.get("SWL_ENGINE_FLAGS", "1"))
        self.wte = weight.wte
        self.reset()

    def repack(self, weight):
        for i, lw in enumerate(weight.layers):
            qkv = lw.qkv_proj if getattr(lw, "qkv_proj", None) is not None else \
                torch.cat((lw.q_proj, lw.k_proj, lw.v_proj), dim=0)
            self.stream[i].copy_(pack_engine_layer(qkv, lw.o_proj, lw.up_gate_proj, lw.down_proj))
            self.norms[i, 0].copy_(lw.attn_norm)
            self.norms[i, 1].copy_(lw.ffn_norm)
        self.wte = weight.wte

    def nbytes(self) -> int:
        return self.stream.numel() * self.stream.element_size()

    def reset(self):
        _hip.call("swl_decode_engine_reset", _hip.ptr(self.ws), self.ws_bytes, _hip.stream())
        self.tok_err.zero_()

    def enable_debug_stamps(self):
        self.debug_stamps = torch.zeros((7, self.cfg.num_layers, 16), dtype=torch.int64, device=self.device)
        return self.debug_stamps

    def step(self, k_cache, v_cache, block_table, input_ids, seq_ids, seq_lens, cos, sin, max_blocks_per_seq: int):
        """Launch the step on torch's current stream; returns the residual stream after the last layer, [1, hidden] (a
        persistent buffer: the next step overwrites it)."""
        cfg = self.cfg
        _hip.call("swl_decode_engine_step", _hip.ptr(self.resid), _hip.ptr(self.stream), _hip.ptr(self.norms),
                  _hip.ptr(self.wte), _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(block_table), _hip.ptr(input_ids),
                  _hip.ptr(seq_ids), _hip.ptr(seq_lens), _hip.ptr(cos), _hip.ptr(sin), _hip.ptr(self.ws), self.ws_bytes,
                  self.tok_err[1:].data_ptr(), _hip.ptr(self.debug_stamps), cfg.num_layers, cfg.hidden_size,
                  cfg.num_q_heads, cfg.num_kv_heads, cfg.head_dim, cfg.ffn_inter_dim, max_blocks_per_seq,
                  cfg.rms_norm_eps, cfg.head_dim ** -0.5, self.flags, _hip.dtype_code(self.dtype), _hip.stream())
        return self.resid
