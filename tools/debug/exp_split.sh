cd $GRAFT_REPO_ROOT
export PLANS="f0 f0 f0 b1 f0 b0;f0 f0 b1 f0 b1 f0 b0;f0 f0 f0 b1 f0 b1;f0 b1 f0 f0 b1 f0 b0;f0 f0 b1 f0 f0 b0"
export CHUNKS="default;64,64,96,32;48,80,96,32;32,96,96,32"
for k in 8 11 14 17 20; do BNHIP_HOST_SPLIT=$k timeout 300 python tools/debug/host_split.py 2>&1 | grep "^split\|Error\|error" ; done > gpurun_out/exp_split2.txt
