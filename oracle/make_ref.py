"""oracle/make_ref.py — stage the REFERENCE's own Python files for the GPU box (TEST INFRASTRUCTURE).

    python -m oracle.make_ref            # build container only: /root/reference must be mounted

/root/reference does not travel to the MI355X box; `oracle/_ref/` does (git-ignored, NOT gpurun-ignored,
exactly like our own built .so files). This script copies the reference's `swiftllm/` package — unmodified
.py files only — into `oracle/_ref/swiftllm/` so that `oracle/ref_triton.py` can JIT the reference's own
Triton kernels with Triton's gfx950 backend on the box: the "Tier 2" oracle of SURVEY.md §8c and the
"reference Triton-path tokens/s" side of the north-star comparison. Nothing under oracle/_ref/ is ever
committed, and nothing in swiftllm_amd/ ever imports it.

Two more things are staged next to it:
  * `oracle/_ref/bf16/swiftllm/` — the same files with every `float16` token replaced by `bfloat16` (torch.float16
    -> torch.bfloat16, tl.float16 -> tl.bfloat16; SURVEY.md H6: the reference hard-codes fp16 at model.py:70,147-148,
    224-225 and in four kernels). The headline dtype is bf16 and the reference has no bf16 path: this mechanical patch
    is the closest thing to "the reference in bf16" (its decode-attention scores then round to bf16 too). The manifest
    records which files changed and how many substitutions each took.
  * `oracle/_ref/examples/` — the reference's own examples/offline.py and examples/online.py, byte for byte, for the
    drop-in acceptance test (tests/test_gpu_reference_examples.py) that runs them against THIS repo's `swiftllm` alias.
"""
import hashlib
import json
import os
import shutil
import sys

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")


def stage(verbose: bool = True) -> str:
    src = os.path.join(REFERENCE, "swiftllm")
    if not os.path.isdir(src):
        raise SystemExit(f"{src} is not mounted: the reference can only be staged in the build container")
    dst = os.path.join(DEST, "swiftllm")
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    manifest = {}
    for root, dirs, files in os.walk(src):
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        for name in sorted(files):
            if not name.endswith(".py"):
                continue
            s = os.path.join(root, name)
            rel = os.path.relpath(s, REFERENCE)
            d = os.path.join(DEST, rel)
            os.makedirs(os.path.dirname(d), exist_ok=True)
            shutil.copyfile(s, d)
            with open(s, "rb") as f:
                manifest[rel] = hashlib.sha256(f.read()).hexdigest()
    # bf16-patched twin
    patched = {}
    bf16_root = os.path.join(DEST, "bf16")
    if os.path.isdir(bf16_root):
        shutil.rmtree(bf16_root)
    for rel in manifest:
        with open(os.path.join(REFERENCE, rel), encoding="utf-8") as f:
            text = f.read()
        n = text.count("float16")
        d = os.path.join(bf16_root, rel)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        with open(d, "w", encoding="utf-8") as f:
            f.write(text.replace("float16", "bfloat16"))
        if n:
            patched[rel] = n
    # the reference's own example scripts, untouched
    examples = {}
    ex_dst = os.path.join(DEST, "examples")
    if os.path.isdir(ex_dst):
        shutil.rmtree(ex_dst)
    os.makedirs(ex_dst)
    for name in ("offline.py", "online.py"):
        s = os.path.join(REFERENCE, "examples", name)
        if os.path.isfile(s):
            shutil.copyfile(s, os.path.join(ex_dst, name))
            with open(s, "rb") as f:
                examples["examples/" + name] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(DEST, "MANIFEST.json"), "w", encoding="utf-8") as f:
        json.dump({"staged_from": REFERENCE, "files": manifest, "bf16_patched_substitutions": patched,
                   "examples": examples}, f, indent=1, sort_keys=True)
    if verbose:
        print(f"[oracle.make_ref] staged {len(manifest)} reference files under {DEST} "
              f"(+ bf16-patched twin: {sum(patched.values())} substitutions in {len(patched)} files; "
              f"{len(examples)} example scripts)")
    return DEST


if __name__ == "__main__":
    stage()
    sys.exit(0)
