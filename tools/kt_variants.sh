#!/bin/bash
# development: kernel-trace stats of one scene for several library variants: tools/kt_variants.sh name W H precision v1 v2 ...
set -o pipefail
scene=$1; w=$2; h=$3; prec=$4; shift 4
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ktv_$v
  mkdir -p $OUT
  export RTC_LIB=$GRAFT_REPO_ROOT/variants/librtc_$v.so
  [ "$v" = "default" ] && unset RTC_LIB
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/tools/one_scene.py $scene $w $h $prec 8 > $OUT/run.log 2>&1 || exit 1
  echo "== $v" >> $GRAFT_REPO_ROOT/gpurun_out/ktv.txt
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $OUT/kt_results.db | grep -v rocclr >> $GRAFT_REPO_ROOT/gpurun_out/ktv.txt
  rm -rf $OUT
done
