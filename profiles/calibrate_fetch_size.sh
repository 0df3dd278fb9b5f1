#!/bin/bash
# What FETCH_SIZE reports for a gather, against the lines the gather is known to touch (profiles/microbench/gather.hip):
# one rocprofv3 --pmc FETCH_SIZE pass per case, the program directly behind `--`.  Prints, per case, the bytes FETCH_SIZE
# reports per load of one 64-byte line by the second (cache-warm) launch.  usage on the GPU box, from the repository root:
#   bash profiles/calibrate_fetch_size.sh > gpurun_out/fetch_size_calibration.txt
set -u
cd $GRAFT_REPO_ROOT/profiles/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/gather gather.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for case in "4 64 1073741824" "8 64 1073741824" "16 64 1073741824" "16 16 1073741824" "16 64 2097152" "16 64 268435456" "16 64 67108864"; do
  rm -rf /tmp/fs
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fs -- /tmp/gather $case > /tmp/fs.log 2>&1 || { echo "case $case failed"; cat /tmp/fs.log | tail -5; continue; }
  python3 - "$case" <<'PY'
import csv, glob, sys
w, lanes, fp = sys.argv[1].split()
rows = []
for f in glob.glob("/tmp/fs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "Gather" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
loads = 512 * 16.0 * int(lanes) * 2048
for i, (d, v) in enumerate(rows):
    print(f"width {w} B/lane, {lanes} lanes, footprint {int(fp) >> 20} MiB, launch {i + 1}: FETCH_SIZE {v:.0f} KiB = {v * 1024 / loads:.1f} bytes per 64-byte line touched ({loads:.3g} loads)")
PY
done
