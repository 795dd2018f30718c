#!/bin/bash
# Per-wave dynamic instruction mix of the fused expand+depthwise kernels (one serial step, depth 1), via rocprofv3 PMC.
#   tools/pmc_expdw.sh <tag> [quick]  -> gpurun_out/<tag>_expdw_pmc.txt   (quick: instruction counts only)
TAG=${1:-r03}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PB="python bench.py --depth 1 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-profile --no-host-pointer --no-secondary --no-distribution"
DBS=""
i=0
CGROUPS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS")
[ "$2" = quick ] && CGROUPS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU")
for c in "${CGROUPS[@]}"; do
  i=$((i + 1)); rm -rf /tmp/pe$i
  rocprofv3 --pmc $c -d /tmp/pe$i -o p -- $PB > /dev/null 2>&1
  DBS="$DBS $(find /tmp/pe$i -name '*.db' | head -1)"
done
LASTN=24 python tools/pmc_layers.py k_expand_dw $DBS > $OUT/${TAG}_expdw_pmc.txt
echo done
