#!/usr/bin/env python
"""Developer tool: one rocprofv3 --pmc pass (counter_collection.csv) -> per-kernel averages of every counter in the pass over the working
launches (launches whose wave-cycle count is below 10 % of the kernel's largest are gated no-ops and left out), plus the derived shares
the design documents quote (parked / issue-stall / issuing shares of the wave cycles; LDS bank-conflict share of the LDS-active cycles).
usage: python tools/sq_summary.py <csv> <label>"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
label = sys.argv[2] if len(sys.argv) > 2 else ""
short = lambda n: re.sub(r"^void ", "", n).split("(")[0].split("<")[0]
disp = collections.OrderedDict()
counters = []
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"k": short(r["Kernel_Name"]), "c": collections.Counter()})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in counters:
        counters.append(r["Counter_Name"])
by = collections.defaultdict(list)
for d in disp.values():
    by[d["k"]].append(d["c"])
derived = []
if {"SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"} <= set(counters):
    derived += [("parked", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ("issue stall", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), ("issuing", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")]
if {"SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"} <= set(counters):
    derived += [("LDS bank-conflict share", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")]
if {"SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"} <= set(counters):
    derived += [("waiting on LDS", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")]
print(f"# SQ counters per launch, {label} (one rocprofv3 --pmc pass, summed over all SQs; working launches only)")
print()
print("Shares: `parked` = SQ_WAIT_ANY (s_waitcnt / barrier), `issue stall` = SQ_WAIT_INST_ANY, `issuing` = SQ_ACTIVE_INST_ANY, each / SQ_WAVE_CYCLES; "
      "`LDS bank-conflict share` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent replaying conflicting lanes of the cycles it was indexing).")
print()
print("| kernel | launches | " + " | ".join(counters) + " | " + " | ".join(d[0] for d in derived) + " |")
print("|---|---|" + "---|" * (len(counters) + len(derived)))
out = []
for k, lst in by.items():
    if k.startswith("at::") or k.startswith("__amd") or k.startswith("void at") or "elementwise" in k:
        continue
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in counters else counters[0]
    big = max(c[key] for c in lst)
    work = [c for c in lst if c[key] >= 0.1 * big] if big > 0 else lst
    m = lambda n: sum(c[n] for c in work) / len(work)
    if m(key) <= 0:
        continue
    cells = [f"{m(c):.3g}" for c in counters] + [(f"{100 * m(a) / m(b):.1f} %" if m(b) > 0 else "-") for _, a, b in derived]
    out.append((m(key), f"| {k} | {len(work)} | " + " | ".join(cells) + " |"))
for _, line in sorted(out, reverse=True):
    print(line)
