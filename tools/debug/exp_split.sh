cd $GRAFT_REPO_ROOT
export PLANS="default"
export CHUNKS="default;32,64,96,64;32,80,80,64;48,64,80,64;48,72,72,64;32,96,64,64;48,80,64,64;56,72,64,64;64,64,80,48;64,72,72,48;48,80,80,48;64,80,64,48"
timeout 300 python tools/debug/host_split.py 2>&1 | grep "^split\|Error\|error" > gpurun_out/exp_split3.txt
