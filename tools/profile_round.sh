#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 ["extra bench.py arguments"]
# -> gpurun_out/<tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of `bench.py` (whole process)
#    gpurun_out/<tag>_kernel_stats_timed.csv   the same trace restricted to the 20 timed steps
#    gpurun_out/<tag>_bench_default.json       the plain bench line (writes the tuning file the other runs read)
#    gpurun_out/<tag>_pmc.csv, <tag>_traffic.json, <tag>_mfma_util.json   PMC passes (one rocprofv3 run per counter group; no tracing flags mixed in)
#    gpurun_out/<tag>_step_detail_depth1.txt   per-launch table of one serial step (bench.py --depth 1 --detail)
#    gpurun_out/<tag>_bench_under_rocprof.json the bench line of the traced run
# Copy what should be judged into profiles/ afterwards.
TAG=${1:-r02}
WL=${2:-}          # extra bench.py arguments, e.g. "--workload perch --precision bf16" (tag the outputs accordingly)
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# one tuning for every process of this script: the first bench run times the candidates and writes the file, every later run
# (kernel trace, each PMC pass, the serial detail run) reads it - so all of them launch the same kernel instantiations
export BNHIP_TUNE_FILE=$OUT/${TAG}_tune.txt
rm -f $BNHIP_TUNE_FILE
python bench.py $WL --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_default.json 2> /dev/null
# what every later pass runs on: the library's source digest, the tune file's hash, the plan signature of the bench line -
# stamped into the counter files so that bench.py quotes them only beside the library they were collected on
python - <<PY
import hashlib, json
line = json.loads(open("$OUT/${TAG}_bench_default.json").read().strip().splitlines()[-1])
ev = line.get("evidence") or {}
json.dump({"lib_digest": ev.get("lib_digest"), "tune_sha256": hashlib.sha256(open("$BNHIP_TUNE_FILE", "rb").read()).hexdigest(),
           "plan_signature": ev.get("plan_signature"), "tag": "$TAG"}, open("$OUT/${TAG}_binding.json", "w"))
PY
BENCH="python bench.py $WL --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-host-pointer --no-secondary --no-distribution"
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/kt.err
DB=$(find /tmp/kt -name "*.db" | head -1)
NL=$(python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_under_rocprof.json").read().strip().splitlines()[-1])
print(sum(k["launches_per_step"] for k in d["kernels_warmup_pass"]))
PY
)
python tools/prof_summary.py $DB --csv $OUT/${TAG}_kernel_stats.csv > /dev/null
python tools/prof_summary.py $DB --csv $OUT/${TAG}_kernel_stats_timed.csv --window $NL:$((20 * NL)) > /dev/null
echo "launches per step: $NL" > $OUT/${TAG}_profile_notes.txt
PB="python bench.py $WL --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-profile --no-host-pointer --no-secondary --no-distribution"
DBS=""
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i + 1)); rm -rf /tmp/pmc$i
  rocprofv3 --pmc $c -d /tmp/pmc$i -o p -- $PB > /dev/null 2>&1
  DB1=$(find /tmp/pmc$i -name '*.db' | head -1)
  if [ -n "$DB1" ]; then DBS="$DBS $DB1"; else echo "pass $i ($c): no database" >> $OUT/${TAG}_profile_notes.txt; fi
done
python tools/pmc_summary.py --traffic-json $OUT/${TAG}_traffic.json --mfma-json $OUT/${TAG}_mfma_util.json --binding $OUT/${TAG}_binding.json --window $NL:$((5 * NL)) $DBS > $OUT/${TAG}_pmc.csv 2> $OUT/${TAG}_pmc_errors.txt
# (the serial detail run is a depth-1 engine: its own tuning, not the pipelined one)
BNHIP_TUNE_FILE= python bench.py $WL --depth 1 --detail --steps 5 --warmup 3 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-host-pointer --no-secondary --no-distribution > /dev/null 2> $OUT/${TAG}_step_detail_depth1.txt
echo done
