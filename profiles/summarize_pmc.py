"""Per kernel: dispatches and per-dispatch means of the rocprofv3 --pmc counters found under <dir>/<pass>/,
plus the kernel-trace duration statistics when <dir>/trace is there.  Writes <dir>/pmc_summary.json and prints
a table.  usage: python summarize_pmc.py <dir>

Derived per-dispatch figures: fabric_read_bytes = FETCH_SIZE (KiB) * 1024 * factor; fabric_write_bytes = WRITE_SIZE (KiB)
* 1024.  Both count requests that leave the L2, including those the Infinity Cache serves.  The factor: on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced streaming read at 64 bytes (factor 2), but reports a
gather of 64-byte lines - one line per lane, 4, 8 or 16 bytes of it used - as it is: 63.8 bytes per line touched over a
1 GiB footprint (profiles/r03_fetch_size_calibration.txt, profiles/calibrate_fetch_size.sh).  The traversal and shading
kernels gather (factor 1; the coalesced part of their traffic - rays in, state of a path - is then under-counted, which
makes their figure a lower bound); the kernels that only stream take factor 2.  valu_utilization = SQ_ACTIVE_INST_VALU
(wave instructions issued, summed over the chip) * 4 cycles per wave64 instruction / (GRBM_GUI_ACTIVE / 8 XCDs,
the kernel's duration in cycles) / 1024 SIMDs: the share of the SIMDs' vector issue slots the kernel uses."""
import csv, glob, json, os, sys
from collections import defaultdict

# kernels whose reads are wide coalesced streams; every other kernel's reads are taken to be gathers
STREAMING = ("RaygenKernel", "FilmKernel", "TonemapKernel", "AddFilmKernel", "ResolveKernel", "FlagKernel", "SwapKernel", "ScanSumsKernel",
             "ScanApplyKernel", "PartnerKernel", "FillItemsKernel", "vectorized_elementwise_kernel")

root = sys.argv[1]
traversal = sys.argv[2] if len(sys.argv) > 2 else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("ptd::", "").replace("void ", "")
    return name.strip()


tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k, c = short(r["Kernel_Name"]), r["Counter_Name"]
        tot[k][c] += float(r["Counter_Value"])
        disp[k][c].add((f, r["Dispatch_Id"]))   # a counter collected in two passes is averaged over both

out = {}
for k in sorted(tot):
    per = {c: tot[k][c] / max(1, len(disp[k][c])) for c in tot[k]}
    n = max(len(v) for v in disp[k].values())
    d = {"dispatches": n, "per_dispatch": per}
    if "FETCH_SIZE" in per:
        d["fetch_size_factor"] = 2 if any(name in k for name in STREAMING) else 1
        d["fabric_read_bytes"] = per["FETCH_SIZE"] * 1024 * d["fetch_size_factor"]
    if "WRITE_SIZE" in per:
        d["fabric_write_bytes"] = per["WRITE_SIZE"] * 1024
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in per and per["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
        d["l1_hit_rate"] = 1 - per.get("TCP_TCC_READ_REQ_sum", 0) / per["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "TCC_REQ_sum" in per and per["TCC_REQ_sum"] > 0:
        d["l2_hit_rate"] = 1 - per.get("TCC_MISS_sum", 0) / per["TCC_REQ_sum"]
    if "SQ_ACTIVE_INST_VALU" in per and per.get("GRBM_GUI_ACTIVE", 0) > 0:
        d["valu_utilization"] = per["SQ_ACTIVE_INST_VALU"] * 4 / (per["GRBM_GUI_ACTIVE"] / 8) / 1024
    if "SQ_INSTS_SALU" in per and per.get("GRBM_GUI_ACTIVE", 0) > 0:
        # one scalar unit per CU, one instruction per cycle at most
        d["salu_utilization"] = per["SQ_INSTS_SALU"] / (per["GRBM_GUI_ACTIVE"] / 8) / 256
    if "TA_TA_BUSY_sum" in per and per.get("GRBM_GUI_ACTIVE", 0) > 0:
        d["ta_busy"] = per["TA_TA_BUSY_sum"] / 256 / (per["GRBM_GUI_ACTIVE"] / 8)   # mean over the 256 texture addressers
    if "SQ_WAVE_CYCLES" in per and per["SQ_WAVE_CYCLES"] > 0:
        d["wave_time_waiting_on_memory"] = per.get("SQ_WAIT_ANY", 0) / per["SQ_WAVE_CYCLES"]
        d["wave_time_issue_stalled"] = per.get("SQ_WAIT_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]
    out[k] = d

for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        out.setdefault(k, {})["trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                                          "percent": float(r["Percentage"])}

# what the counters belong to: bench.py uses a summary only beside the kernels it was collected with
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, repo)
from bench import kernel_source_hash  # noqa: E402  (one list of sources, one hash)
out["_meta"] = {"kernel_sources": kernel_source_hash(), "traversal": traversal}

json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, d in out.items():
    if k == "_meta":
        print(k, d)
        continue
    print(k, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in d.items() if x != "per_dispatch"})
    for c, v in sorted(d.get("per_dispatch", {}).items()):
        print(f"    {c:36s} {v:.6g}")
