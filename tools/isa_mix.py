#!/usr/bin/env python3
"""isa_mix.py — instruction mix and issue order of a kernel's loops, from hipcc's assembly (no GPU needed).

    python tools/isa_mix.py swiftllm_amd/csrc/prefill_attn.hip --kernel 'prefill_attn_kernelIDF16bLi128' [--seq]

Compiles the file to gfx950 assembly, finds the kernel whose mangled name contains --kernel, splits it into basic
blocks, and for every block that is a loop body (a backward branch targets it or it lies between a label and a
backward branch to that label) prints the count of MFMA / VALU / transcendental / LDS / VMEM / SALU / wait
instructions. --seq prints the class of every instruction of the largest loop in issue order (M = MFMA, v = VALU,
t = transcendental, d = LDS, g = global/buffer memory, s = scalar, w = s_waitcnt, b = barrier, n = s_nop): what the
matrix pipe sees between two MFMAs is what decides whether VALU work hides behind it.
"""
import argparse
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "t"
    if op.startswith("v_"):
        return "v"
    if op.startswith("ds_"):
        return "d"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "g"
    if op == "s_waitcnt":
        return "w"
    if op == "s_barrier":
        return "b"
    if op == "s_nop":
        return "n"
    if op.startswith("s_"):
        return "s"
    return "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--seq", action="store_true")
    ap.add_argument("--ops", action="store_true", help="opcode histogram of the largest loop")
    ap.add_argument("-D", dest="defines", action="append", default=[])
    ap.add_argument("--min", type=int, default=40, help="ignore loops with fewer instructions")
    a = ap.parse_args()
    out = os.path.join(tempfile.gettempdir(), "isa_mix.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           f"-I{ROOT}/swiftllm_amd/csrc", "-S", "--cuda-device-only", "-o", out, a.src] + [f"-D{d}" for d in a.defines]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out, encoding="utf-8").read().splitlines()
    start = next(i for i, l in enumerate(lines) if a.kernel in l and l.rstrip().endswith(":") or (a.kernel in l and ": ;" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.endswith(":"):
            continue
        op = s.split()[0]
        insts.append((op, s))
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    meta = [l.strip() for l in lines if any(k in l for k in (".vgpr_count", ".sgpr_count", ".vgpr_spill", ".lds_size", "agpr_count"))]
    print(f"kernel {a.kernel}: {len(insts)} instructions, {len(loops)} loops")
    best = None
    for lo, hi in loops:
        seq = [classify(op) for op, _ in insts[lo:hi + 1]]
        if len(seq) < a.min:
            continue
        cnt = {k: seq.count(k) for k in "MvtdgswbN"}
        cnt["n"] = seq.count("n")
        print(f"  loop [{lo}, {hi}] {len(seq)} instr: " + " ".join(f"{k}={v}" for k, v in cnt.items() if v))
        if best is None or len(seq) > len(best[2]):
            best = (lo, hi, seq)
    if a.ops and best:
        import collections
        hist = collections.Counter(op for op, _ in insts[best[0]:best[1] + 1])
        for op, n in hist.most_common():
            print(f"    {n:4d}  {op}")
    if a.seq and best:
        s = "".join(best[2])
        for i in range(0, len(s), 120):
            print("   ", s[i:i + 120])
        # gaps between consecutive MFMAs
        gaps, cur = [], None
        for c in best[2]:
            if c == "M":
                if cur is not None:
                    gaps.append(cur)
                cur = 0
            elif cur is not None:
                cur += 1
        print("    instructions between consecutive MFMAs:", gaps)


if __name__ == "__main__":
    main()
