# Marginal cost of a kernel class in the PIPELINED step: context 1 skips the class's launches (BNHIP_DEBUG_CTX1_SKIP; its outputs are then
# garbage, the timing is not) - what the bench gains against what the class costs alone says how much of the class the other context hides.
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-secondary --no-distribution --no-host-pointer --no-fp32-run --no-oracle-check"
for v in "" stft expand_dw pw_gemm se "" frontend dwconv clip_minmax mean ""; do
  BNHIP_DEBUG_CTX1_SKIP=$v timeout 300 python bench.py $F --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip[$v]', round(d['value']), round(d['ms_per_step'],4))"
done
