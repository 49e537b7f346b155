"""Small helpers shared by the host code (same names as the reference's swiftllm/utils.py)."""

KB = 1 << 10
MB = 1 << 20
GB = 1 << 30
TB = 1 << 40


def cdiv(a: int, b: int) -> int:
    """Ceiling division for non-negative integers."""
    return -(-a // b)
