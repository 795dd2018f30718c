#!/bin/bash
# Records the create-time tunings of the bench's plans into birdnet-go_amd/tune/ (run on the GPU box through gpurun; copy
# gpurun_out/tune_new/*.tune into birdnet-go_amd/tune/ afterwards).  The tuners are timing races, so TRIES candidate sets are
# recorded and the one whose headline (bench default, 20 steps, twice) is best is kept.   tools/record_tunes.sh [tries=3]
TRIES=${1:-3}
OUT=$PWD/gpurun_out
best=0; bestdir=
for t in $(seq $TRIES); do
  D=/tmp/tune_try$t; rm -rf $D; mkdir -p $D
  BNHIP_TUNE_DIR=$D BNHIP_TUNE_RECORD=1 python bench.py --no-cpu-baseline --no-distribution > /dev/null 2>&1
  BNHIP_TUNE_DIR=$D BNHIP_TUNE_RECORD=1 python bench.py --batch 1024 --steps 6 --no-cpu-baseline --no-secondary --no-host-pointer --no-fp32-run --no-distribution --no-oracle-check > /dev/null 2>&1
  v=0
  for r in 1 2; do
    x=$(BNHIP_TUNE_DIR=$D python bench.py --no-cpu-baseline --no-secondary --no-host-pointer --no-fp32-run --no-distribution --no-oracle-check 2>/dev/null | tail -1 | python -c "import sys,json; print(int(json.loads(sys.stdin.read())['value']))")
    v=$((v + x))
  done
  echo "try $t: $(ls $D | wc -l) plans, headline sum of two runs $v"
  if [ $v -gt $best ]; then best=$v; bestdir=$D; fi
done
rm -rf $OUT/tune_new; mkdir -p $OUT/tune_new; cp $bestdir/*.tune $OUT/tune_new/
echo "kept $bestdir ($best)"; ls $OUT/tune_new
