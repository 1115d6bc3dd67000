"""Developer tool: print the kernel timeline of one forward-only step and one training step from a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'frustum_culling' in r['Kernel_Name']]
want_train = len(sys.argv) > 2 and sys.argv[2] == 'train'
sel = None
for n in range(len(idx) - 1):
    a, b = idx[n], idx[n + 1]
    has_bwd = any('raster_backward' in r['Kernel_Name'] for r in rows[a:b])
    if has_bwd == want_train:
        sel = (a, b)
a, b = sel
t0 = int(rows[a]['Start_Timestamp']); prev = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:5.1f} dur {(e-s)/1e3:7.1f}  {r['Kernel_Name'][:80]}")
    prev = e
print("step total", (int(rows[b]['Start_Timestamp']) - t0) / 1e3, "kernels", b - a)
