#!/usr/bin/env python3
"""make_gemm_wide_ablations.py — the timing-only variants of csrc/gemm_wide.hip that tools/gpu_wide_ablate.sh times (r06d).

Each variant is the shipped source with ONE part of the K loop compiled out (results are WRONG on purpose) or one load hint
changed (`plain`: right results), built as csrc/libswiftllm_hip_gw<name>.so next to the product library (it travels to the
GPU box with the snapshot; delete the files afterwards):

    python tools/make_gemm_wide_ablations.py            # writes /tmp/gw_<name>.hip and builds every variant
    gpurun -- 'bash tools/gpu_wide_ablate.sh'

  plain   W loads without the non-temporal hint in every tiling
  nox     no global loads of x: the staging registers are filled from a register
  nolds   B fragments taken from the weight registers instead of LDS
  wcache  every W ring slot re-reads tile 0 (cache hits instead of the HBM stream)
  nomfma  one fma per fragment pair instead of the MFMA"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "swiftllm_amd", "csrc", "gemm_wide.hip")

EDITS = {
    "plain": [("return TS == 2 ? load8(p) : load8_nt(p);", "return load8(p);")],
    "nox": [("xr[set][q_] = load8(xb + xoff[q_] + (tile) * kWT);",
             "xr[set][q_] = *reinterpret_cast<const vec8_t<T> *>(&xoff[0]);")],
    "nolds": [("bq_[0][mt_] = *reinterpret_cast<const vec8_t<T> *>(xl_ + mt_ * 32 * kWT + ((hf ^ swz) << 3));",
               "bq_[0][mt_] = wr[slot][mt_ % 4]; (void)xl_;"),
              ("bq_[nk_ % BQ][mt_] = *reinterpret_cast<const vec8_t<T> *>(xl_ + mt_ * 32 * kWT + off_);",
               "bq_[nk_ % BQ][mt_] = wr[slot][(nk_ + mt_) % 4]; (void)xl_; (void)off_;")],
    "wcache": [("load_w(wsrc + rb_ * wrb + (static_cast<int64_t>(tile) * 4 + i_) * 512);",
                "load_w(wsrc + rb_ * wrb + (static_cast<int64_t>((tile) & 0) * 4 + i_) * 512);")],
    "nomfma": [("acc[rb_ * MTW + mt_] = mfma_w(wr[slot][rb_ * 4 + kk_], bq_[kk_ % BQ][mt_], acc[rb_ * MTW + mt_]);",
                "acc[rb_ * MTW + mt_][0] += to_f(wr[slot][rb_ * 4 + kk_][0]) * to_f(bq_[kk_ % BQ][mt_][0]);")],
}


def main():
    src = open(SRC).read()
    names = sys.argv[1:] or list(EDITS)
    for name in names:
        text = src
        for old, new in EDITS[name]:
            assert text.count(old) == 1, f"{name}: pattern not found exactly once — the kernel moved on: {old[:60]}"
            text = text.replace(old, new)
        path = f"/tmp/gw_{name}.hip"
        open(path, "w").write(text)
        subprocess.run([sys.executable, "-m", "swiftllm_amd.csrc.build", "--tag", f"gw{name}", "--swap",
                        f"gemm_wide.hip={path}"], cwd=ROOT, check=True)


if __name__ == "__main__":
    main()
