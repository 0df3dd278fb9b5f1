"""Markdown table of the measured configurations: throughput from the clean runs (profiles/<round>_other_configs.jsonl and
the bench line), and per traversal mode the dominant kernel's counters from the committed summaries.
usage: python3 profiles/config_table.py [round tag, default r03]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
HBM_PEAK = 8000.0  # GB/s


def dominant(summary):
    """The kernel with the largest share of the traced time, and its figures per launch"""
    best = None
    for name, k in summary.items():
        if name.startswith("_") or "trace" not in k:
            continue
        if best is None or k["trace"]["total_ns"] > summary[best]["trace"]["total_ns"]:
            best = name
    k = summary[best]
    ms = k["trace"]["avg_ns"] * 1e-6
    fabric = (k.get("fabric_read_bytes", 0) + k.get("fabric_write_bytes", 0)) / (ms * 1e-3) * 1e-9
    units = {"VALU issue": min(1.0, k.get("valu_utilization", 0)), "texture addresser": k.get("ta_busy", 0), "HBM": fabric / HBM_PEAK,
             "scalar unit": k.get("salu_utilization", 0)}
    bound = max(units, key=units.get)
    return best, ms, fabric, bound, units


rows = []
clean = {}
for path in glob.glob(os.path.join(ROOT, f"{tag}*_other_configs.jsonl")):
    for line in open(path):
        if line.startswith("{"):
            r = json.loads(line)
            clean[(r["config"].split()[0], r["traversal"])] = r
for path in glob.glob(os.path.join(ROOT, f"{tag}*_bench_*.json")):
    r = json.loads(open(path).read().strip().splitlines()[-1])
    clean[("atrium", r["roofline"].get("traversal", "wide"))] = {"mrays_per_s": r["value"], "msamples_per_s": r["msamples_per_s"]}

print("| config | traversal | Mrays/s | Msamples/s | dominant kernel | ms / launch | GB/s past the L2 | of 8 TB/s | busiest unit (VALU / TA / HBM) |")
print("|---|---|---|---|---|---|---|---|---|")
for scene, label in (("cornell", "C2 cornell 1024^2, 256 spp"), ("atrium", "C3 atrium 1080p depth 5, 128 spp per pass"),
                     ("knot", "C4 knot 871 k, 4K, primary + shadow, 16 spp"), ("soup", "C5 soup 20 M, 4K depth 5, 8 spp")):
    for mode in ("reference", "ordered", "wide"):
        path = os.path.join(ROOT, f"{scene}_{tag}_{mode}_pmc_summary.json")
        if scene == "atrium":
            path = os.path.join(ROOT, f"{tag}_{mode}_pmc_summary.json")
        if not os.path.exists(path):
            continue
        name, ms, fabric, bound, units = dominant(json.load(open(path)))
        c = clean.get((scene, mode), {})
        print(f"| {label} | {mode} | {c.get('mrays_per_s', '')} | {c.get('msamples_per_s', '')} | {name} | {ms:.2f} | {fabric:.0f} | {fabric / HBM_PEAK:.3f} | "
              f"{bound} ({units['VALU issue']:.2f} / {units['texture addresser']:.2f} / {units['HBM']:.2f}) |")
