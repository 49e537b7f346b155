#!/usr/bin/env python3
"""kres.py <file.hip> [extra hipcc flags] — one line per kernel: VGPR/AGPR/scratch/LDS/occupancy
(from hipcc -Rpass-analysis=kernel-resource-usage, gfx950)."""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra,
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][:80]:80s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} lds={r.get('LDS Size [bytes/block]')} occ={r.get('Occupancy [waves/SIMD]')}")
