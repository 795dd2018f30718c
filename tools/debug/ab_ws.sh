#!/bin/bash
# A/B in the pipelined bench: k_pw_ws as a tuner candidate (default) vs taken away (BNHIP_PW_WS=0), alternating, plus the serial engine
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary --no-host-pointer --no-distribution --no-fp32-run --no-oracle-check --steps 40 --warmup 5"
for i in 1 2; do
  timeout 200 $B > gpurun_out/ab_ws_on_$i.json 2> /dev/null
  BNHIP_PW_WS=0 timeout 200 $B > gpurun_out/ab_ws_off_$i.json 2> /dev/null
done
timeout 200 $B --depth 1 > gpurun_out/ab_ws_on_d1.json 2> /dev/null
BNHIP_PW_WS=0 timeout 200 $B --depth 1 > gpurun_out/ab_ws_off_d1.json 2> /dev/null
tools/ubench/bin/ws_trace > gpurun_out/ws_trace.txt 2>&1
