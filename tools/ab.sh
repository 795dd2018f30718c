#!/bin/bash
# A/B two builds of libbnhip on the SAME GPU box (box-to-box spread is ~6 %): tools/ab.sh [reps]
# expects birdnet-go_amd/lib/libbnhip_A.so (baseline) next to libbnhip.so
R=${1:-2}
for i in $(seq $R); do for L in libbnhip_A.so libbnhip.so; do
  printf "%-16s " $L; BNHIP_LIB=$PWD/birdnet-go_amd/lib/$L python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4))"
done; done
