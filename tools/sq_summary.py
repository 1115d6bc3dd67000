#!/usr/bin/env python
"""Developer tool: one rocprofv3 --pmc SQ pass (counter_collection.csv) -> per-kernel averages over the working launches (launches whose
wave-cycle count is below 10 % of the kernel's largest are gated no-ops and left out).  usage: python tools/sq_summary.py <csv> <label>"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
label = sys.argv[2] if len(sys.argv) > 2 else ""
short = lambda n: re.sub(r"^void ", "", n).split("(")[0]
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"k": short(r["Kernel_Name"]), "c": collections.Counter()})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
by = collections.defaultdict(list)
for d in disp.values():
    by[d["k"]].append(d["c"])
print(f"# SQ counters per launch, {label} (one rocprofv3 --pmc pass, summed over all SQs; working launches only)")
print()
print("`SQ_WAIT_ANY` = wave parked on s_waitcnt/barrier, `SQ_WAIT_INST_ANY` = issue stall, `SQ_ACTIVE_INST_ANY` = issuing (disjoint shares of `SQ_WAVE_CYCLES`; quad-cycles).")
print()
print("| kernel | launches | wave cycles | parked | issue stall | issuing | VALU instructions | VALU busy | SALU instructions |")
print("|---|---|---|---|---|---|---|---|---|")
out = []
for k, lst in by.items():
    if k.startswith("at::") or k.startswith("__amd") or k.startswith("void at") or "elementwise" in k:
        continue
    big = max(c["SQ_WAVE_CYCLES"] for c in lst)
    work = [c for c in lst if c["SQ_WAVE_CYCLES"] >= 0.1 * big] if big > 0 else lst
    m = lambda n: sum(c[n] for c in work) / len(work)
    wc = m("SQ_WAVE_CYCLES")
    if wc <= 0:
        continue
    out.append((wc, f"| {k} | {len(work)} | {wc:.3g} | {100 * m('SQ_WAIT_ANY') / wc:.0f} % | {100 * m('SQ_WAIT_INST_ANY') / wc:.0f} % | "
                    f"{100 * m('SQ_ACTIVE_INST_ANY') / wc:.0f} % | {m('SQ_INSTS_VALU'):.3g} | {m('SQ_ACTIVE_INST_VALU'):.3g} | {m('SQ_INSTS_SALU'):.3g} |"))
for _, line in sorted(out, reverse=True):
    print(line)
