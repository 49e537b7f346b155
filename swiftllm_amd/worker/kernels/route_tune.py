"""Which kernel serves a decode projection of 65..256 tokens: the hand-written `swl_gemm_packed_wide` (csrc/gemm_wide.hip)
or the library GEMM (`F.linear` -> hipBLASLt, the reference's own call: swiftllm/worker/kernels/linear.py:3-12)?

Both compute the same product (one rounding of an fp32-accumulated sum) but sum K in different orders, so WHICH one runs
is part of a deployment's numerics: it must not depend on a timing race inside a request, and every replica of a
data-parallel deployment must answer alike.

  * the measured table (r04 sweep, qkv / o re-measured with cold weights in r06d: Llama-3-8B widths, bfloat16, ROCm 7.2)
    answers for the (N, K, dtype) classes it was measured on (`MEASURED`);
  * any other class is measured ONCE PER DEPLOYMENT, at `LlamaModel.load_weights()` time — before any request and before
    any hipGraph capture — for every 32-token bucket of 65..256 tokens (`prepare`, driven by kernels/linear.py:
    tune_wide_routes): a few launches of each side, the hand-written kernel taken when it wins by more than 3 %. The table
    is written next to the checkpoint (`<model_path>/swiftllm_amd_routes.json`; `SWIFTLLM_ROUTE_CACHE` overrides, and a
    read-only model directory falls back to ~/.cache/swiftllm_amd/) under a lock file, so the first replica of a node
    measures and every other replica READS the same answer;
  * `decide()` is a pure lookup: a class nobody measured goes to the library (deterministic), it never times anything,
    takes no lock and asks the driver for nothing;
  * `SWIFTLLM_ROUTE_TUNE=table` pins the measured table for every shape (tests that assert WHICH kernel ran), `=off` sends
    every unmeasured shape to the library without measuring."""
import json
import os
import time

import torch

MEASURED_DTYPE = torch.bfloat16
# (N, K) of the Llama-3-8B projections the r04 sweep covered: fused qkv, o_proj, up/gate, down
MEASURED = {(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)}
_TOKEN_BUCKET = 32
BUCKETS = tuple(range(96, 257, _TOKEN_BUCKET))      # 65..96, ..128, ... ..256 tokens
_WIN_MARGIN = 1.03
_LOCK_WAIT_S = 120.0

_table = {}             # "N,K,dtype,bucket,silu" -> bool: this process's deployment table
_device_key_cache = {}


def table_wide_wins(m: int, n: int, k: int) -> bool:
    """Llama-3-8B widths, bf16, MI355X, us ours / library. r04 sweep (profiles/r04c_ / r04d_gemm_wide_micro.jsonl; M = 96, 128,
    160, 192, 224, 256):
        down (K >= 2N)   33/63  34/75  43/85  44/104  52/60  54/63      -> always
    qkv and o re-measured in r06d (profiles/r06d_gemm_wide_routing_remeasure.jsonl) with the weights cycled through 1.2 GB —
    the r04 sweep's six copies left 200-300 MB of them in the Infinity Cache, which flattered the library's 160-token
    kernels (o: 24 us then, 31.5 cold) — and with the token-split tiling of csrc/gemm_wide.hip; M = 136, 144, 160, 176, 192,
    208, 224, 256:
        qkv  (N = 6144)  27/41  27/41  28/31  29/30  29/37  37/30  37/40  38/43   -> always
        o    (N = 4096)  22/28  22/24  22/32  23/26  24/36  29/27  29/29  30/27   -> up to 192
    (up to 128 tokens both were already ours.) Thresholds sit on the 32-token bucket edges of the hipGraph replay cache
    (`LlamaModel._decode_batch_bucket`) and nowhere else: a batch of 200 runs as 200 rows eagerly and as 224 rows in a replayed
    graph, and both must take the same side — the library's one good qkv kernel at 208 tokens is therefore not used (the
    bucket's replay shape, 224, is ours: `test_large_decode_batches_replay_their_hip_graph_bit_for_bit`). The plain up/gate
    projection only ties (53 / 54 at 128) and loses beyond; its SiLU-gate form is `table_wide_silu_wins`."""
    if k >= 2 * n:
        return True
    if n > 8192:
        return False
    return m <= 192 or n > 4096


def table_wide_silu_wins(m: int) -> bool:
    """up/gate projection + SiLU-gate in one launch against the library GEMM + silu_and_mul: 53.5 / 63.3 us at 128 tokens,
    60.0 / 62.5 at 96; r06d (pipelined B fragments, cold weights): 81 / 71 at 136, 77 / 71 at 160, 81 / 76 at 192, 108 / 87 at
    256: up to 128 tokens."""
    return m <= 128


def _mode() -> str:
    return os.environ.get("SWIFTLLM_ROUTE_TUNE", "auto")


def _key(n: int, k: int, dtype, bucket: int, silu: bool) -> str:
    return f"{n},{k},{str(dtype).replace('torch.', '')},{bucket},{int(silu)}"


def bucket_of(m: int) -> int:
    return -(-m // _TOKEN_BUCKET) * _TOKEN_BUCKET


def is_measured_class(n: int, k: int, dtype) -> bool:
    return dtype == MEASURED_DTYPE and (n, k) in MEASURED


def decide(m: int, n: int, k: int, dtype: torch.dtype, silu: bool) -> bool:
    """True = the hand-written kernel. A pure lookup (see the module docstring)."""
    if _mode() == "table" or is_measured_class(n, k, dtype):
        return table_wide_silu_wins(m) if silu else table_wide_wins(m, n, k)
    return bool(_table.get(_key(n, k, dtype, bucket_of(m), silu), False))


# ---- the once-per-deployment measurement ------------------------------------------------------------------------------------
def device_key(device) -> str:
    key = str(device)
    if key not in _device_key_cache:
        try:
            name = torch.cuda.get_device_name(device)
        except Exception:     # noqa: BLE001
            name = "unknown"
        _device_key_cache[key] = f"{name}|hip {getattr(torch.version, 'hip', None)}|torch {torch.__version__}"
    return _device_key_cache[key]


def table_paths(model_path: str):
    """Where the deployment's table may live, in order of preference."""
    env = os.environ.get("SWIFTLLM_ROUTE_CACHE")
    if env:
        return [env]
    out = []
    if model_path and os.path.isdir(model_path):
        out.append(os.path.join(model_path, "swiftllm_amd_routes.json"))
    out.append(os.path.join(os.path.expanduser("~"), ".cache", "swiftllm_amd", "routes.json"))
    return out


def _read(path: str) -> dict:
    try:
        with open(path, encoding="utf-8") as f:
            data = json.load(f)
        return data if isinstance(data, dict) else {}
    except (OSError, ValueError):
        return {}


def _write(path: str, data: dict) -> bool:
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(data, f, indent=0, sort_keys=True)
        os.replace(tmp, path)
        return True
    except OSError:
        return False


def time_us(fn, iters: int = 8, warm: int = 2) -> float:
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def prepare(classes, dtype, device, model_path: str, measure) -> dict:
    """Fill this process's table for `classes` — an iterable of (N, K, silu) — at every token bucket.
    `measure(n, k, silu, bucket) -> bool` times the two candidates (kernels/linear.py supplies it). Classes the r04 table
    answers are skipped; answers already in the deployment's file (same device + software key) are READ, not re-measured;
    what is missing is measured under a lock file so that concurrently starting replicas do it once. Returns
    {"measured": n, "read": n, "path": file or None}."""
    stats = {"measured": 0, "read": 0, "path": None}
    if _mode() in ("table", "off"):
        return stats
    want = [(n, k, bool(silu), b) for (n, k, silu) in sorted(set(classes)) if not is_measured_class(n, k, dtype)
            for b in BUCKETS]
    if not want:
        return stats
    dkey = device_key(device)
    paths = table_paths(model_path)

    def load_known():
        for p in paths:
            dev = _read(p).get(dkey)
            if isinstance(dev, dict) and all(_key(n, k, dtype, b, s) in dev for n, k, s, b in want):
                return p, dev
        return None, None
    path, dev = load_known()
    lock = None
    if dev is None:
        # one measurer per node: whoever creates the lock file measures, the others wait for the table to appear
        for p in paths:
            try:
                os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
                fd = os.open(p + ".lock", os.O_CREAT | os.O_EXCL | os.O_WRONLY)
                os.close(fd)
                lock, path = p + ".lock", p
                break
            except FileExistsError:
                deadline = time.monotonic() + _LOCK_WAIT_S
                while time.monotonic() < deadline:
                    path, dev = load_known()
                    if dev is not None or not os.path.exists(p + ".lock"):
                        break
                    time.sleep(0.2)
                if dev is not None:
                    break
                path = p            # the measurer died or timed out: measure ourselves (same code, same device)
                break
            except OSError:
                continue            # read-only location: try the next one
    if dev is not None:
        for n, k, s, b in want:
            _table[_key(n, k, dtype, b, s)] = bool(dev[_key(n, k, dtype, b, s)])
        stats.update(read=len(want), path=path)
        return stats
    try:
        fresh = {}
        for n, k, s, b in want:
            try:
                fresh[_key(n, k, dtype, b, s)] = bool(measure(n, k, s, b))
            except Exception:     # noqa: BLE001 — a shape our kernel refuses: the library serves it
                fresh[_key(n, k, dtype, b, s)] = False
        _table.update(fresh)
        stats.update(measured=len(want))
        if path is not None:
            data = _read(path)
            data.setdefault(dkey, {}).update(fresh)
            if _write(path, data):
                stats["path"] = path
    finally:
        if lock is not None:
            try:
                os.remove(lock)
            except OSError:
                pass
    return stats


def wins(ours_us: float, library_us: float) -> bool:
    return bool(ours_us * _WIN_MARGIN < library_us)
