"""Developer tool: per-kernel time per executor step from a rocprofv3 kernel trace of bench.py (steps are delimited by the chunk
culling launch of the native executor; only steps that ran the fused backward+Adam kernel are kept; the first `skip` are dropped)."""
import collections
import csv
import sys

path, skip = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60]))
rows.sort()
steps, cur = [], None
for s, e, n in rows:
    if n.startswith("frustum_culling_chain"):
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((s, e, n))
steps = [st for st in steps if any(n.startswith("project_backward_adam") for _, _, n in st)
         and not any(n.startswith("activate_forward") or n.startswith("radix_rank_selftest") for _, _, n in st)][skip:]
per, calls = collections.defaultdict(float), collections.defaultdict(int)
wall = 0.0
for st in steps:
    for s, e, n in st:
        per[n] += (e - s) / 1e3
        calls[n] += 1
    wall += (st[-1][1] - st[0][0]) / 1e3
k = len(steps)
print(f"{k} steps; kernel-time sum {sum(per.values()) / k:.1f} us/step; first-start..last-end {wall / k:.1f} us/step")
for n in sorted(per, key=lambda x: -per[x]):
    print(f"{n:62s} {per[n] / k:8.1f} us/step  {calls[n] / k:5.2f} launches/step  {per[n] / calls[n]:8.1f} us avg")
