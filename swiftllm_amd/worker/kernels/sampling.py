"""Greedy sampling. Reference: swiftllm/worker/layers/post_layer.py:40 (`torch.argmax(logits, dim=1)`)."""
import torch

from swiftllm_amd import _hip

_scratch = {}   # device -> persistent candidate buffer (fixed address: hipGraph replays use it)
_retired = []   # outgrown buffers stay allocated: a captured hipGraph may still replay against them


def argmax_rows(logits: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """[rows, n] fp16/bf16 -> int64 [rows]; ties go to the lowest index (what torch.argmax does too). `out`: a caller-owned
    contiguous int64 [rows] destination (the one-sequence decode engine keeps its error word next to the token)."""
    rows, n = logits.shape
    if n % 8 or logits.stride(1) != 1 or logits.stride(0) % 8 or rows > 65535 or \
            logits.dtype not in (torch.float16, torch.bfloat16):
        res = torch.argmax(logits, dim=1)       # odd vocabularies: the generic device reduce
        return res if out is None else out.copy_(res)
    need = _hip.load().swl_argmax_scratch_bytes(rows)
    buf = _scratch.get(logits.device)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(need, 512 * 64 * 8), dtype=torch.uint8, device=logits.device)
        _scratch[logits.device] = buf
    if out is None:
        out = torch.empty((rows,), dtype=torch.int64, device=logits.device)
    assert out.dtype == torch.int64 and out.is_contiguous() and out.numel() == rows
    _hip.call("swl_argmax", _hip.ptr(out), _hip.ptr(logits), _hip.ptr(buf), buf.numel(), rows, n,
              logits.stride(0), _hip.dtype_code(logits.dtype), _hip.stream())
    return out
