cd $GRAFT_REPO_ROOT
{ timeout 300 python tools/debug/host_mid.py 2>&1 | grep pipe_min
BNHIP_HOST_NOSPLIT=1 timeout 300 python tools/debug/host_mid.py 2>&1 | grep pipe_min
for r in 16 32; do BNHIP_HOST_PIPE_MIN=32 BNHIP_HOST_RAMP=$r timeout 300 python tools/debug/host_mid.py 2>&1 | grep pipe_min; done
BNHIP_HOST_PIPE_MIN=32 BNHIP_HOST_RAMP=32 BNHIP_HOST_NOSPLIT=1 timeout 300 python tools/debug/host_mid.py 2>&1 | grep pipe_min
BNHIP_HOST_PIPE_MIN=64 timeout 300 python tools/debug/host_mid.py 2>&1 | grep pipe_min; } > gpurun_out/exp_mid.txt
