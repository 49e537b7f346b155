#!/bin/bash
# tools/gpu_pmc_rows.sh — r05: HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of
# the SiLU-gate GEMM with norm on the fly (the dominant kernel of the r05 decode step) and of the row-owned o_proj / down_proj.
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/pmc_rows; mkdir -p $O; cd /tmp
run() { # name, kernel substring, command...
  name=$1; pat=$2; shift 2
  "$@" 2>/dev/null | tail -1 > $O/$name.micro.json
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/$name.$c
    rocprofv3 --pmc $c --kernel-trace -d $O/$name.$c -o p -- "$@" > $O/$name.$c.log 2>&1
    python $R/tools/rocpd_pmc.py $(find $O/$name.$c -name "*.db" | head -1) $pat > $O/$name.$c.txt
    rm -rf $O/$name.$c
  done
  python - $O $name "$*" <<'PY'
import json, re, sys
o, name, cmd = sys.argv[1:4]
def counter(c):
    t = open(f"{o}/{name}.{c}.txt").read()
    m = re.search(rf"{c}: dispatches=(\d+) mean=([\d.]+).*mean_duration_us=([\d.]+)", t)
    return int(m.group(1)), float(m.group(2)), float(m.group(3))
nf, fetch_kb, dur = counter("FETCH_SIZE")
nw, write_kb, _ = counter("WRITE_SIZE")
micro = json.loads(open(f"{o}/{name}.micro.json").read())
alg = micro.get("algorithmic_bytes")
traffic = 2 * fetch_kb * 1024 + write_kb * 1024
out = dict(kernel=micro.get("kernel"), command=f"rocprofv3 --pmc FETCH_SIZE --kernel-trace -- {cmd}  (and a separate pass with --pmc WRITE_SIZE); tools/gpu_pmc_rows.sh",
           micro=micro, dispatches=nf, FETCH_SIZE_KB_mean=fetch_kb, WRITE_SIZE_KB_mean=write_kb,
           fetch_correction="x2: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced 16-B/lane streams (MI355X_MICROARCH.md, HBM section)",
           traffic_bytes_per_launch=int(traffic), algorithmic_bytes_per_launch=int(alg),
           traffic_over_algorithmic=round(traffic / alg, 4), mean_duration_us_under_profiler=dur)
json.dump(out, open(f"{o}/{name}_pmc.json", "w"), indent=1)
print(name, json.dumps({k: out[k] for k in ("traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic", "mean_duration_us_under_profiler")}))
PY
}
run gemm_silu_nf gemm_skinny_ring python $R/tools/gemm_silu_micro.py --nf
run gemm_rows_o gemm_rows_kernel python $R/tools/gemm_rows_micro.py --pmc o --m 32 --iters 64 --copies 16
run gemm_rows_down gemm_rows_kernel python $R/tools/gemm_rows_micro.py --pmc down --m 8 --iters 64 --copies 8
