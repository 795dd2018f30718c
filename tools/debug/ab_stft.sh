#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_frontend_forms.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/stft_test.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-host-pointer --no-distribution --no-fp32-run --no-oracle-check --steps 40 --warmup 5"
for i in 1 2; do
  timeout 200 $B > gpurun_out/ab_prune_on_$i.json 2> /dev/null
  BNHIP_STFT_PRUNE=0 timeout 200 $B > gpurun_out/ab_prune_off_$i.json 2> /dev/null
done
timeout 200 $B --depth 1 --detail --steps 5 > /dev/null 2> gpurun_out/ab_prune_on_detail.txt
BNHIP_STFT_PRUNE=0 timeout 200 $B --depth 1 --detail --steps 5 > /dev/null 2> gpurun_out/ab_prune_off_detail.txt
