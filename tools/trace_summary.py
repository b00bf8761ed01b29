"""Summarise a rocprofv3 kernel trace of a steady-state loop: per-kernel time per step, busy time and idle gaps.

    python tools/trace_summary.py <kernel_trace.csv> <steps> [skip_fraction]

The trace of the last `steps` generations is located by taking the final (1 - skip_fraction) share of the dispatches
(default: the last half), so warm-up and plan timing do not enter the table."""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]
span = rows[-1][1] - rows[0][0]
busy = 0
cur_s, cur_e = rows[0][0], rows[0][1]
per = defaultdict(lambda: [0, 0])
for s, e, name in rows:
    per[name][0] += e - s
    per[name][1] += 1
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
frac = 1.0 - skip
n_steps = steps * frac
print(f"dispatches {len(rows)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us ({100*busy/span:.1f}%)  idle {(span-busy)/1e3:.1f} us")
print(f"per step (assuming {n_steps:.0f} steps in the window): span {span/1e3/n_steps:.1f} us, busy {busy/1e3/n_steps:.1f} us, "
      f"kernels {len(rows)/n_steps:.1f}")
tot = sum(v[0] for v in per.values())
for name, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:28]:
    short = name if len(name) < 90 else name[:87] + "..."
    print(f"{t/1e3/n_steps:8.1f} us/step  {c/n_steps:5.2f} x {t/c/1e3:7.1f} us  {100*t/tot:5.1f}%  {short}")
