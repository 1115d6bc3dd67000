#!/usr/bin/env python
"""Developer tool: gpurun_out/{prof_TAG, pmc_FETCH_SIZE, pmc_WRITE_SIZE, pmc_sqA} -> profiles/rNN_* (tracked).

  profiles/rNN_step_timeline.md   one training step of the executor, kernel by kernel (averages over the timed steps of the trace)
  profiles/rNN_pmc_summary.md     HBM-side traffic per launch (FETCH_SIZE / WRITE_SIZE passes) against the algorithmic bytes
  profiles/rNN_hbm_traffic.json   the same numbers, machine readable
  profiles/rNN_sq_counters.md     issue counters per kernel

The executor enqueues a gated second copy of eight kernels every step (the fallback of the depth-bound culling); those launches exit
at once.  They are separated from the working launches by their value (< 10 % of the kernel's largest launch) and reported apart.
usage: python tools/profile_summary.py TAG [round]"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else "r02"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def short(name):
    return re.sub(r"^void ", "", name).split("(")[0]


def timeline():
    f = glob.glob(os.path.join(G, f"prof_{TAG}", "*kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    idx = [i for i, n in enumerate(names) if n.startswith("frustum_culling_chain")]
    acc, nsteps, falls, tot_steps = collections.OrderedDict(), 0, 0, []
    for s in range(len(idx) - 1):
        a, b = idx[s], idx[s + 1]
        if not any("project_backward_adam" in n for n in names[a:b]):
            continue                                       # forward-only steps of the bench (fwd Msplats/s) are not training steps
        if s < 24:
            continue                                       # setup + warm-up
        dur = lambda i: (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        fell = sum(1 for i in range(a, b) if names[i].startswith("raster_forward") and dur(i) > 20) > 1
        falls += fell
        tot_steps.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
        if fell:
            continue                                       # steps whose culled run had to be repeated unculled are summarised separately
        nsteps += 1
        cnt = collections.Counter()
        for i in range(a, b):
            k = (names[i], cnt[names[i]])
            cnt[names[i]] += 1
            acc[k] = acc.get(k, 0.0) + dur(i)
    out = [f"# Round {RND[1:]}: one training step of the native executor, 3 M Gaussians @1080p (rocprofv3 --kernel-trace, `bench.py --steps 40 --warmup 16`)", "",
           f"{len(tot_steps)} timed training steps in the trace, mean {sum(tot_steps) / len(tot_steps):.1f} us start to start; {falls} of them repeated the frame unculled "
           f"(depth bound violated: the gated fallback ran).  The table averages the other {nsteps}.  Durations are dispatch to completion: "
           "each includes the ~5 us of a dependent launch, which is ALL a gated fallback launch (`#1` rows) costs when it is not needed.", "",
           "| kernel | launch | us |", "|---|---|---|"]
    total = 0.0
    for (n, k), v in acc.items():
        out.append(f"| {n} | #{k} | {v / nsteps:.1f} |")
        total += v / nsteps
    out += [f"| **sum** | | **{total:.1f}** |", ""]
    open(os.path.join(P, f"{RND}_step_timeline.md"), "w").write("\n".join(out))
    print("\n".join(out[:6]), f"\n... sum {total:.1f} us")


def per_kernel(counter, sub):
    f = os.path.join(G, sub, "r02_counter_collection.csv")
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter:
            vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    out = {}
    for k, v in vals.items():
        big = max(v)
        main = [x for x in v if x >= 0.1 * big] if big > 0 else v
        out[k] = (sum(main) / len(main), len(main), len(v) - len(main))
    return out


# algorithmic bytes per launch at the bench workload, for the comparison column
def algorithmic(units):
    N, I, Iv, Ic, Pp = units["n_vis"], units["I"], units["I_vis"], units["I_c"], units["P"]
    px = 1920 * 1080 * 3
    return {"project_fused_kernel<3, 8, 16>": None, "raster_backward_kernel<8, 16, false, false, false>": Pp * 18 + Iv * 36 + Ic * 36,
            "raster_forward_kernel<8, 16, false>": Iv * 36 + Pp * 18, "l1_ssim_forward_kernel": px * 20, "l1_ssim_backward_kernel": px * 24,
            "radix_onesweep_kernel<2, false>": 16 * units["emitted"], "radix_onesweep_kernel<1, false>": 16 * N, "tile_range_kernel": 4 * units["emitted"]}


def pmc():
    F, W = per_kernel("FETCH_SIZE", "pmc_FETCH_SIZE"), per_kernel("WRITE_SIZE", "pmc_WRITE_SIZE")
    bench = json.loads(open(os.path.join(G, f"bench_{TAG}.log")).read().strip().splitlines()[-1])
    units = dict(bench["roofline"]["units_per_launch"])
    units["emitted"] = bench["instances"]
    alg = algorithmic(units)
    out = [f"# Round {RND[1:]} PMC summary (rocprofv3 --pmc, one counter per pass, bench.py 3m_1080p, per-launch averages of the working launches)", "",
           "FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced",
           "stream, other widths and WRITE_SIZE are uncalibrated, and Infinity-Cache hits appear to be counted.  Two figures are therefore given: the raw",
           "`F + W` and the guide's corrected `2F + W` (an upper bound for kernels that read 4 B/lane, e.g. the SSIM pair, whose raw fetch already matches",
           "their halo-inclusive reads: 32x32 tiles with a 5-pixel apron re-read (42/32)^2 = 1.72x of every map that is blurred).", "",
           "| kernel | working launches (gated no-ops) | FETCH KiB | WRITE KiB | F+W MB | 2F+W MB | algorithmic MB |", "|---|---|---|---|---|---|---|"]
    traffic, rows = {}, []
    for k, (f, n, gated) in F.items():
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        w = W.get(k, (0.0, 0, 0))[0]
        raw, cor = (f + w) * 1024, (2 * f + w) * 1024
        traffic[k] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "raw_bytes": int(raw), "corrected_bytes": int(cor)}
        rows.append((cor, k, f, w, raw, n, gated))
    for cor, k, f, w, raw, n, gated in sorted(rows, reverse=True):
        a = alg.get(k)
        out.append(f"| {k} | {n} ({gated}) | {f:.0f} | {w:.0f} | {raw / 1e6:.1f} | {cor / 1e6:.1f} | {'' if a is None else f'{a / 1e6:.1f}'} |")
    out.append("")
    open(os.path.join(P, f"{RND}_pmc_summary.md"), "w").write("\n".join(out))
    json.dump({"commit_note": "profiles of the build that produced profiles/%s_bench_3m.json" % RND, "units_per_launch": units, "kernels": traffic},
              open(os.path.join(P, f"{RND}_hbm_traffic.json"), "w"), indent=1)
    print("\n".join(out[8:22]))


def sq():
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"]
    C = {n: per_kernel(n, "pmc_sqA") for n in names}
    out = [f"# Round {RND[1:]} SQ counters (one rocprofv3 --pmc pass, bench.py 3m_1080p, per-launch averages of the working launches, summed over all SQs)", "",
           "`SQ_WAIT_ANY` = wave parked on s_waitcnt/barrier, `SQ_WAIT_INST_ANY` = issue stall, `SQ_ACTIVE_INST_ANY` = issuing (disjoint shares of",
           "`SQ_WAVE_CYCLES`).", "",
           "| kernel | wave cycles | parked | issue stall | issuing | VALU instructions | SALU instructions |", "|---|---|---|---|---|---|---|"]
    for k, (wc, n, g) in sorted(C["SQ_WAVE_CYCLES"].items(), key=lambda kv: -kv[1][0]):
        if k.startswith("at::") or k.startswith("__amd") or wc <= 0:
            continue
        get = lambda c: C[c].get(k, (0.0, 0, 0))[0]
        out.append(f"| {k} | {wc:.3g} | {100 * get('SQ_WAIT_ANY') / wc:.0f} % | {100 * get('SQ_WAIT_INST_ANY') / wc:.0f} % | {100 * get('SQ_ACTIVE_INST_ANY') / wc:.0f} % | "
                   f"{get('SQ_INSTS_VALU'):.3g} | {get('SQ_INSTS_SALU'):.3g} |")
    out.append("")
    open(os.path.join(P, f"{RND}_sq_counters.md"), "w").write("\n".join(out))
    print("\n".join(out[5:14]))


if __name__ == "__main__":
    timeline()
    pmc()
    sq()
