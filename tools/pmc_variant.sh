#!/bin/bash
# development: memory-pipe counters of one scene for library variants: tools/pmc_variant.sh scene W H precision v1 v2 ...
set -o pipefail
scene=$1; w=$2; h=$3; prec=$4; shift 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  OUT=$R/gpurun_out/pmcv_$v
  rm -rf $OUT; mkdir -p $OUT
  export RTC_LIB=$R/variants/librtc_$v.so
  [ "$v" = "default" ] && unset RTC_LIB
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES TD_TD_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $OUT -o pmc -- python3 $R/tools/one_scene.py $scene $w $h $prec 4 > $OUT/run.log 2>&1 || exit 1
  echo "== $v $scene $prec" >> $R/gpurun_out/pmcv.txt
  python3 $R/tools/pmc_dump.py $OUT/pmc_results.db 1 >> $R/gpurun_out/pmcv.txt
  rm -rf $OUT
done
