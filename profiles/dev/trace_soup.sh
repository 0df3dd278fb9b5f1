cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/profiles/render_config.py soup --spp 8 --traversal wide > /tmp/tr.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/tr/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"])>0.5: print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "max us", round(float(r["MaxNs"])/1e3,1), "%", r["Percentage"])
PY
