#!/usr/bin/env python
"""Developer tool: rocprofv3 kernel trace of `tools/late_phase.py trace` -> per-kernel averages over the LAST n training steps of the trace
(a step = frustum_culling_chain ... project_backward_adam; everything before -- teacher renders, the statistics pass, first visits -- is
left out).  usage: python tools/late_timeline.py <kernel_trace.csv> [n]"""
import collections, csv, re, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
short = lambda s: re.sub(r"^void ", "", s).split("(")[0]
names = [short(r["Kernel_Name"]) for r in rows]
dur = lambda i: (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
starts = [i for i, k in enumerate(names) if k.startswith("frustum_culling_chain")]
steps = []
for s, lo in enumerate(starts):
    hi = starts[s + 1] if s + 1 < len(starts) else len(rows)
    last = max((i for i in range(lo, hi) if "project_backward_adam" in names[i]), default=None)
    if last is not None:
        steps.append(list(range(lo, last + 1)))
steps = steps[-n:]
acc, order = collections.defaultdict(float), []
wall = 0.0
for st in steps:
    cnt = collections.Counter()
    for i in st:
        k = (names[i].split("<")[0] if names[i].startswith("at::") else names[i], cnt[names[i]]); cnt[names[i]] += 1
        if k not in acc:
            order.append(k)
        acc[k] += dur(i)
    wall += (int(rows[st[-1]]["End_Timestamp"]) - int(rows[st[0]]["Start_Timestamp"])) / 1e3
print(f"# Late-phase step timeline: average over the last {len(steps)} training steps of the trace (us per step, dispatch to completion)")
print()
tot = sum(acc.values()) / max(len(steps), 1)
print(f"kernel sum {tot:.0f} us per step; first dispatch to last completion {wall / max(len(steps), 1):.0f} us per step")
print()
print("| kernel (# = n-th launch of that kernel in the step) | us / step | share |")
print("|---|---:|---:|")
for k in order:
    v = acc[k] / len(steps)
    if v >= 0.5:
        print(f"| {k[0]}{' #%d' % k[1] if k[1] else ''} | {v:.1f} | {100 * v / tot:.1f} % |")
