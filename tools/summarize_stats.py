#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats (rocpd sqlite) -> the small per-kernel text summary committed under profiles/.
usage: summarize_stats.py <results.db> <out.txt> "<header line>" """
import sqlite3, sys
db, out, header = sys.argv[1], sys.argv[2], sys.argv[3]
con = sqlite3.connect(db)
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
con.close()
lines = ["# " + header, "# columns: calls  total_us  avg_us  pct  kernel"]
for name, calls, tot, avg, pct in rows:
    lines.append(f"{calls:6d} {tot:12.1f} {avg:10.3f} {pct:6.2f}  {name[:170]}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
