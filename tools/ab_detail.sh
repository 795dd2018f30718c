#!/bin/bash
# A/B of two builds on the SAME box, layer by layer: serial (depth 1) per-launch table of one kernel class, both builds on ONE recorded
# tuning (the first run writes it), alternating runs.   tools/ab_detail.sh [class=expand_dw] [reps=2]
# expects birdnet-go_amd/lib/libbnhip_A.so (baseline) next to libbnhip.so
K=${1:-expand_dw}; R=${2:-2}
export BNHIP_TUNE_FILE=/tmp/ab_tune.txt; rm -f $BNHIP_TUNE_FILE
A="--depth 1 --detail --steps 5 --warmup 3 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-host-pointer --no-secondary --no-distribution"
for i in $(seq $R); do for L in libbnhip_A.so libbnhip.so; do
  echo "== $L run $i"; BNHIP_LIB=$PWD/birdnet-go_amd/lib/$L python bench.py $A 2>&1 >/dev/null | grep " $K " | awk '{printf "%-22s %8s us\n", $3, $4; t+=$4} END {printf "%-22s %8.1f us\n", "TOTAL", t}'
done; done
