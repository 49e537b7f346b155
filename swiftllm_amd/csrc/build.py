"""Build libswiftllm_hip.so (gfx950) in-tree with hipcc. No torch headers, no cmake: seconds.

    python -m swiftllm_amd.csrc.build [--force] [--jobs N]

The library lands next to the sources (swiftllm_amd/csrc/libswiftllm_hip.so), is git-ignored,
and travels to the GPU box with the repo snapshot.
"""
import argparse
import concurrent.futures
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = [
    "rmsnorm.hip",
    "rotary.hip",
    "kvcache.hip",
    "silu_mul.hip",
    "paged_attn.hip",
    "prefill_attn.hip",
    "block_table.hip",
    "swap_blocks.hip",
    "gemm_skinny.hip",
    "gemm_tiny.hip",
    "gemm_wide.hip",
    "gemm_rows.hip",
    "argmax.hip",
    "decode_engine.hip",
]
HEADERS = ["swl_common.h", "attend_block.h", os.path.join(ROOT, "include", "swiftllm_hip.h")]
LIB = os.path.join(HERE, "libswiftllm_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
# -ffp-contract=off: the reference's rounding points (fp16 mul, then fp16 add) must not be fused.
CXXFLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
            "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, jobs: int = 0, verbose: bool = True, tag: str = "", defines=(), swap=None) -> str:
    """Build the library. `tag` + `defines` (-D macros) / `swap` build an EXPERIMENT variant next to it
    (libswiftllm_hip_<tag>.so; its objects go to the system temp directory, not into the tree that ships to the GPU
    box): tools/ select one with SWIFTLLM_HIP_LIB to A/B a kernel on the GPU box without touching the product library."""
    hipcc = _hipcc()
    obj_dir = os.path.join(tempfile.gettempdir(), "swiftllm_hip_variants", tag) if tag else OBJ_DIR
    lib = os.path.join(HERE, f"libswiftllm_hip_{tag}.so") if tag else LIB
    return _build(hipcc, obj_dir, lib, [f"-D{d}" for d in defines], force, jobs, verbose, swap or {})


def _build(hipcc, OBJ_DIR, LIB, extra, force, jobs, verbose, swap) -> str:
    """`swap`: {source name: replacement path} — an experiment variant may compile another file in a source's place
    (e.g. an older revision of one kernel for an A/B run)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    hdrs.append(os.path.abspath(__file__))
    todo, objs = [], []
    for src in SOURCES:
        src_path = swap.get(src, os.path.join(HERE, src))
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _newer(obj, [src_path] + hdrs):
            todo.append((src_path, obj))

    def compile_one(item):
        src_path, obj = item
        cmd = [hipcc, *CXXFLAGS, *extra, f"-I{HERE}", "-c", src_path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src_path}:\n{r.stderr}")
        return src_path

    if todo:
        jobs = jobs or min(len(todo), os.cpu_count() or 4)
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            for done in ex.map(compile_one, todo):
                if verbose:
                    print(f"[swiftllm_amd.csrc] compiled {os.path.basename(done)}", flush=True)
    if todo or force or not _newer(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[swiftllm_amd.csrc] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--tag", default="", help="build an experiment variant libswiftllm_hip_<tag>.so")
    ap.add_argument("-D", dest="defines", action="append", default=[], help="macro for the variant")
    ap.add_argument("--swap", action="append", default=[], metavar="NAME=PATH", help="compile PATH in place of source NAME")
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, tag=a.tag, defines=a.defines, swap=dict(x.split("=", 1) for x in a.swap)))
