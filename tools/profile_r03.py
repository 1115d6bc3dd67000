#!/usr/bin/env python
"""Developer tool: one rocprofv3 kernel trace (and optionally one --pmc SQ pass) of the default bench.py run -> per-kernel averages of one
training step in the TWO states bench.py reports: the fresh cloud (the timed steps) and the trained-state cloud (the 100 steps after the
soak).  Training steps are found by their first kernel (frustum_culling_chain) and last (project_backward_adam).

usage: python tools/profile_r03.py <kernel_trace.csv> [<counter_collection.csv>] [--setup 8 --warmup 16 --steps 40 --probe 16 --soak 1000]
writes markdown to stdout."""
import argparse, collections, csv, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("counters", nargs="?")
ap.add_argument("--setup", type=int, default=8); ap.add_argument("--warmup", type=int, default=16); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--probe", type=int, default=16); ap.add_argument("--soak", type=int, default=1000)
a = ap.parse_args()


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0]


def segment(rows, key):
    """rows sorted in dispatch order -> list of training steps, each a list of row indices"""
    names = [short(r["Kernel_Name"]) for r in rows]
    starts = [i for i, n in enumerate(names) if n.startswith("frustum_culling_chain")]
    steps = []
    for s in range(len(starts) - 1):
        lo, hi = starts[s], starts[s + 1]
        idx = [i for i in range(lo, hi)]
        last = max((i for i in idx if "project_backward_adam" in names[i]), default=None)
        if last is None:
            continue                                   # forward-only passes
        steps.append([i for i in idx if i <= last])
    return names, steps


fresh_lo = a.setup + a.warmup
fresh = range(fresh_lo, fresh_lo + a.steps)
steady_lo = fresh_lo + a.steps + a.probe + a.soak
steady = range(steady_lo, steady_lo + 100)

rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
names, steps = segment(rows, None)
dur = lambda i: (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3


def table(sel):
    acc, n = collections.OrderedDict(), 0
    span = []
    for s in sel:
        if s >= len(steps):
            continue
        st = steps[s]
        if sum(1 for i in st if names[i].startswith("raster_forward") and dur(i) > 20) > 1:
            continue                                   # the frame was repeated unculled: summarised apart
        n += 1
        cnt = collections.Counter()
        for i in st:
            k = (names[i], cnt[names[i]]); cnt[names[i]] += 1
            acc[k] = acc.get(k, 0.0) + dur(i)
        span.append((int(rows[st[-1]]["End_Timestamp"]) - int(rows[st[0]]["Start_Timestamp"])) / 1e3)
    return acc, n, (sum(span) / len(span) if span else 0.0)


fa, fn, fspan = table(fresh)
sa, sn, sspan = table(steady)
print(f"# One training step of the native executor, kernel by kernel, in both states bench.py reports (rocprofv3 --kernel-trace of the default run)")
print()
print(f"{len(steps)} training steps in the trace.  Fresh cloud: steps {fresh.start}..{fresh.stop - 1} ({fn} averaged, first dispatch to last completion "
      f"{fspan:.1f} us).  Trained-state cloud: steps {steady.start}..{steady.stop - 1} ({sn} averaged, {sspan:.1f} us).  Steps that repeated the frame unculled are left out.")
print("Durations are dispatch to completion (each includes ~5 us of dependent-launch latency).")
print()
print("| kernel | launch | fresh us | trained us |")
print("|---|---|---|---|")
keys = list(fa.keys()) + [k for k in sa.keys() if k not in fa]
tf = ts = 0.0
for k in keys:
    f = fa.get(k, 0.0) / max(fn, 1); s = sa.get(k, 0.0) / max(sn, 1)
    tf += f; ts += s
    print(f"| {k[0]} | #{k[1]} | {f:.1f} | {s:.1f} |")
print(f"| **sum** | | **{tf:.1f}** | **{ts:.1f}** |")

if a.counters:
    crow = sorted(csv.DictReader(open(a.counters)), key=lambda r: (int(r["Dispatch_Id"]), r["Counter_Name"]))
    # one row per (dispatch, counter): regroup per dispatch
    disp = collections.OrderedDict()
    for r in crow:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    drows = [{"Kernel_Name": v["Kernel_Name"], **v["c"]} for v in disp.values()]
    cnames, csteps = segment(drows, None)

    def ctable(sel):
        acc, cnt = {}, collections.Counter()
        for s in sel:
            if s >= len(csteps):
                continue
            for i in csteps[s]:
                k = cnames[i]
                big = drows[i].get("SQ_WAVE_CYCLES", 0.0)
                if big < 1e5:
                    continue                           # gated no-op launches
                cnt[k] += 1
                for c, v in drows[i].items():
                    if c != "Kernel_Name":
                        acc.setdefault(k, collections.Counter())[c] += v
        return {k: {c: v / cnt[k] for c, v in acc[k].items()} for k in acc}

    for label, sel in (("fresh cloud", fresh), ("trained-state cloud", steady)):
        t = ctable(sel)
        print()
        print(f"## SQ counters per launch, {label} (one --pmc pass; summed over all SQs; working launches only)")
        print()
        print("| kernel | wave cycles | parked | issue stall | issuing | VALU instructions | VALU busy cycles | SALU instructions |")
        print("|---|---|---|---|---|---|---|---|")
        for k, c in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            if wc <= 0 or k.startswith("at::") or k.startswith("__amd"):
                continue
            g = lambda n: c.get(n, 0.0)
            print(f"| {k} | {wc:.3g} | {100 * g('SQ_WAIT_ANY') / wc:.0f} % | {100 * g('SQ_WAIT_INST_ANY') / wc:.0f} % | {100 * g('SQ_ACTIVE_INST_ANY') / wc:.0f} % | "
                  f"{g('SQ_INSTS_VALU'):.3g} | {g('SQ_ACTIVE_INST_VALU'):.3g} | {g('SQ_INSTS_SALU'):.3g} |")
