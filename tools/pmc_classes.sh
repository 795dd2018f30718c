#!/bin/bash
# Dynamic instruction CLASS mix per wave of the fused expand + depthwise kernels (one serial step, depth 1), via rocprofv3 PMC:
# VALU split into transcendental / fp32 mul / add / fma / int32 / int64 / cvt (the SQ_INSTS_VALU_* counters; packed ops count once),
# MFMA, LDS, VMEM, SALU - what tools/isa_audit.py --classes prints statically, here as executed.
#   tools/pmc_classes.sh <tag> [kernel-substring]  -> gpurun_out/<tag>_expdw_classes.txt
TAG=${1:-r06}
PAT=${2:-k_expand_dw}
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PB="python bench.py --depth 1 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-run --no-oracle-check --no-profile --no-host-pointer --no-secondary --no-distribution"
DBS=""
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" "SQ_WAVES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_WAVES SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i + 1)); rm -rf /tmp/pc$i
  rocprofv3 --pmc $c -d /tmp/pc$i -o p -- $PB > /dev/null 2>&1
  DBS="$DBS $(find /tmp/pc$i -name '*.db' | head -1)"
done
LASTN=24 python tools/pmc_layers.py $PAT $DBS > $OUT/${TAG}_expdw_classes.txt
echo done
