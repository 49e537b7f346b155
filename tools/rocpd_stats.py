#!/usr/bin/env python3
"""rocpd_stats.py <rocprofv3 results .db> [top N] — per-kernel summary (calls, total, mean, min, max, %)
from the SQLite database rocprofv3 (ROCm 7.2) writes for --kernel-trace. Prints a markdown table."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"| kernel | calls | total ms | mean us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for name, n, tot, avg, mn, mx in rows[:top]:
    short = re.sub(r"\(.*", "", name)
    short = short if len(short) <= 90 else short[:87] + "..."
    print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
print(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
