#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel summary committed
under profiles/.  usage: summarize_rocpd.py <results.db> [<out.md>]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % | vgpr | sgpr | lds B | scratch B | grid_x | wg_x |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    name = r[0].split("(")[0]
    lines.append(f"| {name} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e6:.4f} | {r[4]/1e6:.4f} | {r[5]/1e6:.4f} | {100*r[2]/total:.2f} | "
                 f"{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
pmc = db.execute("select count(*) from pmc_events").fetchone()[0]
if pmc:
    lines.append("")
    lines.append("| kernel | counter | dispatches | mean value per dispatch |")
    lines.append("|---|---|---|---|")
    try:
        q = ("select k.name, p.counter_name, count(*), avg(p.counter_value) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name order by k.name")
        for r in db.execute(q):
            lines.append(f"| {r[0].split('(')[0]} | {r[1]} | {r[2]} | {r[3]:.1f} |")
    except Exception as e:  # schema differences between rocprofv3 builds
        lines.append(f"| (pmc query failed: {e}) | | | |")
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(out)
print(out)
