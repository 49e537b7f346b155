"""oracle/make_ref.py — stage the REFERENCE's own Python files for the GPU box (TEST INFRASTRUCTURE).

    python -m oracle.make_ref            # build container only: /root/reference must be mounted

/root/reference does not travel to the MI355X box; `oracle/_ref/` does (git-ignored, NOT gpurun-ignored,
exactly like our own built .so files). This script copies the reference's `swiftllm/` package — unmodified
.py files only — into `oracle/_ref/swiftllm/` so that `oracle/ref_triton.py` can JIT the reference's own
Triton kernels with Triton's gfx950 backend on the box: the "Tier 2" oracle of SURVEY.md §8c and the
"reference Triton-path tokens/s" side of the north-star comparison. Nothing under oracle/_ref/ is ever
committed, and nothing in swiftllm_amd/ ever imports it.
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
    with open(os.path.join(DEST, "MANIFEST.json"), "w", encoding="utf-8") as f:
        json.dump({"staged_from": REFERENCE, "files": manifest}, f, indent=1, sort_keys=True)
    if verbose:
        print(f"[oracle.make_ref] staged {len(manifest)} reference files under {DEST}")
    return DEST


if __name__ == "__main__":
    stage()
    sys.exit(0)
