#!/bin/bash
# development: cache counters of the dragons frame, flattened against instanced (RTC_INSTANCING overrides compile_scene's choice)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/pmc_inst.txt
for prec in fp64 fp32; do
for inst in 0 1; do
  OUT=$R/gpurun_out/pmci_$inst
  rm -rf $OUT; mkdir -p $OUT
  export RTC_INSTANCING=$inst
  timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT -o pmc -- python3 $R/tools/one_scene.py dragons 1920 1080 $prec 10 > $OUT/run.log 2>&1 || exit 1
  echo "== dragons 1920x1080 $prec instancing $inst" >> $R/gpurun_out/pmc_inst.txt
  python3 $R/tools/pmc_dump.py $OUT/pmc_results.db 1 >> $R/gpurun_out/pmc_inst.txt
  rm -rf $OUT
done
done
