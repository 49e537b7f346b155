#!/usr/bin/env python3
"""rocpd_pmc.py <results.db> <kernel substring> — per-dispatch mean of every collected counter for the
kernels whose name contains the substring (rocprofv3 --pmc ... --kernel-trace, ROCm 7.2 SQLite output)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = f"%{sys.argv[2]}%"
rows = db.execute("select counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                  "from counters_collection where kernel_name like ? group by counter_name", (pat,)).fetchall()
for name, n, avg, mn, mx, dur in rows:
    print(f"{name}: dispatches={n} mean={avg:.1f} min={mn:.1f} max={mx:.1f} mean_duration_us={dur / 1e3:.2f}")
