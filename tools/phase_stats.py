"""Developer tool: per-kernel average duration per phase of tools/cull_ab.py from a rocprofv3 kernel trace (steps are delimited by the
frustum culling launch; phases are runs of 48 steps: ON, OFF, ON, OFF)."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').split('(')[0][:46]))
rows.sort()
step = -1
per = [collections.defaultdict(float) for _ in range(4)]
cnt = [0] * 4
for s, e, n in rows:
    if n.startswith('frustum_culling_chain'):
        step += 1
    if step < 0:
        continue
    ph = step // 48
    if ph > 3 or step % 48 < 16:          # skip each phase's warm-up steps
        continue
    per[ph][n] += (e - s) / 1e3
    if n.startswith('frustum_culling_chain'):
        cnt[ph] += 1
names = sorted(set().union(*[set(p) for p in per]), key=lambda k: -per[1].get(k, 0))
print(f"{'kernel':48s} " + "  ".join(f"{'ON' if i % 2 == 0 else 'OFF':>8s}" for i in range(4)))
tot = [0.0] * 4
for n in names:
    vals = [per[i][n] / max(cnt[i], 1) for i in range(4)]
    for i in range(4):
        tot[i] += vals[i]
    if max(vals) > 1.0:
        print(f"{n:48s} " + "  ".join(f"{v:8.1f}" for v in vals))
print(f"{'sum per step (us)':48s} " + "  ".join(f"{v:8.1f}" for v in tot))
