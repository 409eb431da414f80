#!/bin/bash
# kernel-trace of one scene render loop (development aid). Usage: tools/kt_scene.sh scene W H precision
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_scene
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/tools/one_scene.py $1 $2 $3 $4 mega 6 > /dev/null 2> $OUT/err.txt || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $OUT/kt_results.db
