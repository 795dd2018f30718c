#!/bin/bash
# one GPU call: bit-identity of k_pw_ws (forced) + what the tuner measures for it as a candidate + the bench with / without it
cd $GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_perch_like.py -m gpu -x -q -p no:cacheprovider -k weights_in_lds 2>&1 | grep -E "passed|failed|rror|assert|FAILED" | tail -6 > gpurun_out/ws_test.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-host-pointer --no-distribution --no-fp32-run --steps 20 --warmup 3"
BNHIP_PW_WS=1 BNHIP_DEBUG=1 timeout 50 $B > gpurun_out/ws_on.json 2> gpurun_out/ws_on.err
grep -E "tune (b1[3-6]/expand|top) .*\(bf16x3\)" gpurun_out/ws_on.err | grep -E "n=256" > gpurun_out/ws_tune.txt
timeout 40 $B > gpurun_out/ws_off.json 2> /dev/null
