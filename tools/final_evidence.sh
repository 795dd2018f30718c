#!/bin/bash
# Round-end evidence in one GPU call (run through gpurun from the repo root): tests -m gpu, the profile sets of both workloads,
# the driver's own bench command, the configs[2] shard size on one GPU, host-pointer rates, the distributed path on one GPU (RCCL with
# one rank; two gloo ranks sharing the GPU for the per-rank ingest leg), the two-shard handle, the real-time tick, the fused
# kernels' instruction-class mix.
#   tools/final_evidence.sh r06
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 > $O/${TAG}_gputests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/${TAG}_gputests_tail.txt 2>&1
timeout 1500 tools/profile_round.sh $TAG > /dev/null 2>&1
timeout 1500 tools/profile_round.sh ${TAG}_perch_bf16 "--workload perch --precision bf16" > /dev/null 2>&1
timeout 900 python bench.py > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err
timeout 600 python bench.py --batch 1024 --no-cpu-baseline --no-secondary --no-host-pointer > $O/${TAG}_bench_b1024.json 2> /dev/null
timeout 600 python tools/hostrate.py --json $O/${TAG}_hostrate_final.json > $O/${TAG}_hostrate_final.txt 2>&1
BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-host-pointer --no-distribution > $O/${TAG}_bench_force_dist_nccl.json 2> /dev/null
BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 6 --warmup 2 > $O/${TAG}_two_ranks_one_gpu.json 2> /dev/null
timeout 600 python tools/debug/two_shards.py > $O/${TAG}_two_shards.json 2> /dev/null
timeout 300 python tools/debug/tick_trace.py > $O/${TAG}_tick_trace.txt 2>&1
timeout 300 python tools/latency_small.py 1 > $O/${TAG}_latency_small_calls.txt 2>&1
timeout 300 python tools/latency_small.py 8 2>&1 | head -3 >> $O/${TAG}_latency_small_calls.txt
timeout 900 tools/pmc_classes.sh $TAG > /dev/null 2>&1
echo done
