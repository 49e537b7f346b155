"""Layout checks shared by the attention-side operators.

The reference asserts `.is_contiguous()` on q/k/v; here a [tokens, heads, head_dim] tensor only has
to be dense in its last two dims — the token pitch is free — so q, k and v may be column slices of
one fused qkv projection output.
"""
import torch


def token_stride(t: torch.Tensor, name: str) -> int:
    """Validate a [tokens, heads, head_dim] view and return its token pitch in elements."""
    assert t.dim() == 3, f"{name} must be [tokens, heads, head_dim]"
    heads, dim = t.shape[1], t.shape[2]
    assert t.stride(2) == 1 and (heads == 1 or t.stride(1) == dim), \
        f"{name} must be dense in its (heads, head_dim) dims, got strides {t.stride()}"
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), heads * dim)
