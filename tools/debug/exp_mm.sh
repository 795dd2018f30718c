cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-secondary --no-distribution --no-host-pointer --no-fp32-run --no-oracle-check"
for v in 0 1 2 3 4 0 2 3; do
  BNHIP_MM_FORM=$v timeout 300 python bench.py $F --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('form $v', round(d['value']), round(d['ms_per_step'],4), d.get('consistent'), d.get('max_abs_logit_diff_vs_small_batch'))"
done
