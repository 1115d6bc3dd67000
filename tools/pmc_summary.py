"""Developer tool: turn rocprofv3 --pmc passes (gpurun_out/pmc_<COUNTER>/r01_counter_collection.csv) into
profiles/r01_pmc_summary.md and profiles/r01_hbm_traffic.json (per-launch averages per kernel)."""
import csv, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out")
counters = sys.argv[1:] or ["FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]

def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

per = {}
for c in counters:
    f = os.path.join(base, f"pmc_{c}", "r01_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != c:
            continue
        k = short(row["Kernel_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    per[c] = {k: v[0] / v[1] for k, v in acc.items()}

out = ["# Round 1 PMC summary (rocprofv3 --pmc, one counter per pass, bench.py 3m_1080p, per-launch averages)", "",
       "FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3). MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of a wide",
       "coalesced stream, so `hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024` is used as the traffic figure (upper bound for",
       "non-streaming patterns; WRITE_SIZE uncalibrated).", ""]
traffic = {}
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    out += ["| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | traffic MB (2F+W) |", "|---|---|---|---|"]
    rows = []
    for k in per["FETCH_SIZE"]:
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        F, Wr = per["FETCH_SIZE"][k], per["WRITE_SIZE"].get(k, 0.0)
        t = (2 * F + Wr) * 1024
        traffic[k] = int(t)
        rows.append((t, k, F, Wr))
    for t, k, F, Wr in sorted(rows, reverse=True):
        out.append(f"| {k} | {F:.0f} | {Wr:.0f} | {t/1e6:.1f} |")
    out.append("")
if "SQ_LDS_BANK_CONFLICT" in per:
    out += ["| kernel | SQ_LDS_BANK_CONFLICT (cycles) | SQ_LDS_IDX_ACTIVE (cycles) | conflict share |", "|---|---|---|---|"]
    for k, v in sorted(per["SQ_LDS_BANK_CONFLICT"].items(), key=lambda kv: -kv[1]):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        a = per.get("SQ_LDS_IDX_ACTIVE", {}).get(k, 0.0)
        out.append(f"| {k} | {v:.3g} | {a:.3g} | {v/a if a else 0:.2f} |")
    out.append("")
open(os.path.join(ROOT, "profiles", "r01_pmc_summary.md"), "w").write("\n".join(out))
if traffic:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json"), "w"), indent=1)
print("\n".join(out[:40]))
