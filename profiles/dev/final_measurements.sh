#!/bin/bash
# On the GPU box, from the repository root: everything the round's tables are made from, into gpurun_out/<tag>_*.
# usage: profiles/dev/final_measurements.sh <tag>
set -u
TAG=${1:-r03}
OUT=gpurun_out
for m in wide ordered reference; do
  timeout -k 10 420 profiles/collect_pmc.sh ${TAG}_$m $m > /dev/null 2>&1 || { echo "collect $m failed"; exit 1; }
  # the bench lines below take their counters from the summary of these very sources
  cp $OUT/${TAG}_$m/pmc_summary.json profiles/${TAG}_${m}_pmc_summary.json
done
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench_wide.json 2> $OUT/${TAG}_bench_wide.err || { echo "bench failed"; exit 1; }
for m in ordered reference; do
  timeout -k 10 300 python3 bench.py --traversal $m --cpu-baseline-spp 0 > $OUT/${TAG}_bench_$m.json 2> $OUT/${TAG}_bench_$m.err || { echo "bench $m failed"; exit 1; }
done
timeout -k 10 900 python3 profiles/other_configs.py --traversal all > $OUT/${TAG}_other_configs.jsonl 2> $OUT/${TAG}_other_configs.err || { echo "other configs failed"; exit 1; }
for m in wide ordered reference; do
  timeout -k 10 300 profiles/collect_pmc.sh ${TAG}_knot_$m profiles/render_config.py knot --spp 16 --traversal $m > /dev/null 2>&1 || { echo "knot $m failed"; exit 1; }
  timeout -k 10 300 profiles/collect_pmc.sh ${TAG}_cornell_$m profiles/render_config.py cornell --spp 64 --traversal $m > /dev/null 2>&1 || { echo "cornell $m failed"; exit 1; }
done
for m in wide ordered; do
  timeout -k 10 500 profiles/collect_pmc.sh ${TAG}_soup_$m profiles/render_config.py soup --spp 4 --traversal $m > /dev/null 2>&1 || { echo "soup $m failed"; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/microbench/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 60 /tmp/valu_rate > $OUT/${TAG}_valu_rate.txt 2>&1
# the configuration the metric is quoted on, in full, through the command line program, and the 20 M triangle soup
(cd pathtracer_amd && timeout -k 10 300 ./pathTracer --generate atrium --width 1920 --height 1080 --spp 1024 --stats --output /tmp/atrium.jpg) > $OUT/${TAG}_cli.log 2>&1 || echo "cli atrium failed"
(cd pathtracer_amd && timeout -k 10 300 ./pathTracer --generate soup --spp 8 --stats --output /tmp/soup.jpg) > $OUT/${TAG}_cli_soup.log 2>&1 || echo "cli soup failed"
tail -c 600 $OUT/${TAG}_bench_wide.json
