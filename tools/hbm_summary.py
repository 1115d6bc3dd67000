#!/usr/bin/env python
"""Developer tool: per-kernel HBM-side traffic against the chip's roofline from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do
not fit one pass) that were collected WITH --kernel-trace, so every dispatch carries its own duration.

    python tools/hbm_summary.py <fetch counter_collection.csv> <unused> <write counter_collection.csv> <label> [last K dispatches per kernel]

Per kernel, averages over the working launches (launches below 10 % of the kernel's largest fetch are gated no-ops): duration, FETCH / WRITE
in MB, the guide's gfx950 correction 2F + W (MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read; an upper bound for
kernels that read 4 B per lane) next to the raw F + W, and both as GB/s and as a fraction of the 8 TB/s HBM3E peak.  Durations are those
of the FETCH pass (counter collection serialises dispatches: they are in-kernel times, a few percent above the un-instrumented trace)."""
import collections, csv, re, sys

fetch_csv, trace_csv, write_csv = sys.argv[1:4]
label = sys.argv[4] if len(sys.argv) > 4 else ""
LAST = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # > 0: only the last LAST dispatches of every kernel (the steady tail of a run that first has to reach its state)
short = lambda n: re.sub(r"^void ", "", n).split("(")[0]
PEAK = 8.0e12


def per_dispatch(path, counter):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = out.setdefault(int(r["Dispatch_Id"]), [short(r["Kernel_Name"]), 0.0])
        d[1] += float(r["Counter_Value"])
    return out


dur = {}
for r in csv.DictReader(open(fetch_csv)):               # the counter file carries every dispatch's own timestamps
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
F = per_dispatch(fetch_csv, "FETCH_SIZE")
Wd = per_dispatch(write_csv, "WRITE_SIZE")
byk_f, byk_w, byk_t = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
for did, (k, v) in F.items():
    byk_f[k].append(v); byk_t[k].append(dur.get(did, 0.0))
for did, (k, v) in Wd.items():
    byk_w[k].append(v)
print(f"# HBM-side traffic per kernel launch, {label}")
print()
print("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one pass each; KiB counters -> MB), working launches only.  `2F+W`: the guide's gfx950 "
      "correction (FETCH_SIZE counts half of a wide coalesced read) -- an upper bound for 4-byte-per-lane readers; `F+W`: raw.  Fractions are of 8 TB/s.")
print()
print("| kernel | launches | us | fetch MB | write MB | F+W MB | 2F+W MB | F+W GB/s (frac) | 2F+W GB/s (frac) |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
rows = []
for k in byk_f:
    if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "rocprim" in k:
        continue
    f, t = byk_f[k], byk_t[k]
    w = byk_w.get(k, [0.0])
    if LAST > 0:
        f, t, w = f[-LAST:], t[-LAST:], w[-LAST:]
    big = max(max(f), 1e-9)
    keep = [i for i, v in enumerate(f) if v >= 0.1 * big]
    bigw = max(max(w), 1e-9)
    wk = [v for v in w if v >= 0.1 * bigw] or [0.0]
    fm = sum(f[i] for i in keep) / len(keep) * 1024 / 1e6
    tm = sum(t[i] for i in keep) / len(keep)
    wm = sum(wk) / len(wk) * 1024 / 1e6
    if tm <= 0 or fm + wm < 1.0:
        continue
    raw, cor = fm + wm, 2 * fm + wm
    rows.append((tm, f"| {k} | {len(keep)} | {tm:.1f} | {fm:.1f} | {wm:.1f} | {raw:.1f} | {cor:.1f} | {raw / tm * 1e3:.0f} ({raw * 1e6 / (tm * 1e-6) / PEAK:.2f}) | {cor / tm * 1e3:.0f} ({cor * 1e6 / (tm * 1e-6) / PEAK:.2f}) |"))
for _, line in sorted(rows, reverse=True):
    print(line)
