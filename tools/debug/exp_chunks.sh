cd $GRAFT_REPO_ROOT
for mb in 256 128 64; do BNHIP_HOST_RAMP=64 timeout 300 python tools/debug/chunk_plan.py $mb 2>&1 | grep max_batch; done > gpurun_out/exp_chunk_plan.txt
BNHIP_HOST_RAMP=128 timeout 300 python tools/debug/chunk_plan.py 128 2>&1 | grep max_batch >> gpurun_out/exp_chunk_plan.txt
BNHIP_HOST_RAMP=32 timeout 300 python tools/debug/chunk_plan.py 64 2>&1 | grep max_batch >> gpurun_out/exp_chunk_plan.txt
