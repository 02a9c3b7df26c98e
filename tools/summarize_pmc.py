"""Folds the rocprofv3 --pmc passes of tools/profile.sh into one JSON: per-dispatch means of every counter for
the dominant kernel (the decode kernel: the one with the largest total duration in the kernel trace)."""
import csv, glob, json, sys
from collections import defaultdict

tag, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
stats = list(csv.DictReader(open(f"{out}/{tag}_kernel_stats.csv")))
dom = max(stats, key=lambda r: float(r["TotalDurationNs"]))
kname = dom["Name"]
sums, cnts = defaultdict(float), defaultdict(int)
for f in glob.glob(f"{out}/pmc*/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)          # (dispatch, counter) -> value summed over its dimensions (XCDs, SEs)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] != kname: continue
        per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per_dispatch.items():
        sums[c] += v; cnts[c] += 1
mean = {c: sums[c] / cnts[c] for c in sorted(sums)}
res = {"kernel": kname, "bench_args": args, "dispatches": {c: cnts[c] for c in sorted(cnts)},
       "kernel_trace_avg_ns": float(dom["AverageNs"]), "per_dispatch_mean": mean,
       "notes": "separate --pmc passes (tools/profile.sh); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* count quad-cycles; "
                "FETCH_SIZE/WRITE_SIZE count KB"}
if "FETCH_SIZE" in mean:
    # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section): the
    # read side is doubled before it is compared with a byte count.  Cross-check on this kernel family: a decode
    # kernel that reads each frame's LLRs exactly once reports half of the known first-touch bytes raw.
    res["hbm_bytes_per_launch"] = {"FETCH_SIZE_raw_bytes": mean["FETCH_SIZE"] * 1024,
                                   "FETCH_SIZE_corrected_bytes": 2 * mean["FETCH_SIZE"] * 1024,
                                   "WRITE_SIZE_bytes": mean.get("WRITE_SIZE", 0) * 1024}
json.dump(res, open(f"{out}/{tag}_pmc.json", "w"), indent=1)
w = mean.get("SQ_WAVES", 0)
print(json.dumps({k: res[k] for k in ("kernel", "kernel_trace_avg_ns")}), file=sys.stderr)
if w:
    occ = mean["SQ_WAVE_CYCLES"] * 4 / (mean["GRBM_GUI_ACTIVE"] / 8) / 256   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    print(f"per wave: VALU {mean['SQ_INSTS_VALU'] / w:.0f}  SALU {mean['SQ_INSTS_SALU'] / w:.0f}  LDS {mean['SQ_INSTS_LDS'] / w:.0f}  "
          f"occupancy {occ:.1f} waves/CU", file=sys.stderr)
