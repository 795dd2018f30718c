#!/bin/bash
# Round-end evidence in one GPU call (run through gpurun from the repo root): tests -m gpu, the profile sets of both workloads,
# the driver's own bench command, host-pointer rates, the distributed path on one GPU, the two-shard handle, the GEMM lab.
#   tools/final_evidence.sh r05
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 > $O/${TAG}_gputests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/${TAG}_gputests_tail.txt 2>&1
timeout 1500 tools/profile_round.sh $TAG > /dev/null 2>&1
timeout 1500 tools/profile_round.sh ${TAG}_perch_bf16 "--workload perch --precision bf16" > /dev/null 2>&1
timeout 900 python bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err
timeout 600 python tools/hostrate.py --json $O/${TAG}_hostrate_final.json > $O/${TAG}_hostrate_final.txt 2>&1
BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-host-pointer --no-distribution > $O/${TAG}_bench_force_dist_nccl.json 2> /dev/null
timeout 600 python tools/debug/two_shards.py > $O/${TAG}_two_shards.json 2> /dev/null
timeout 300 tools/ubench/bin/pw_lab > $O/${TAG}_pw_lab_final.txt 2>&1
timeout 120 tools/ubench/bin/ws_trace > $O/${TAG}_ws_trace.txt 2>&1
echo done
