cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PB="python bench.py --depth 1 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-profile"
DBS=""
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i + 1)); rm -rf /tmp/pq$i
  rocprofv3 --pmc $c -d /tmp/pq$i -o p -- $PB > /dev/null 2>&1
  DBS="$DBS $(find /tmp/pq$i -name '*.db' | head -1)"
done
for d in $DBS; do LASTN=12 python tools/pmc_layers.py k_expand_dw $d; echo; done
