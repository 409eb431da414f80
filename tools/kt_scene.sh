#!/bin/bash
# kernel-trace stats of one scene on the GPU box: tools/kt_scene.sh name W H precision tag
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- python3 $GRAFT_REPO_ROOT/tools/one_scene.py $1 $2 $3 $4 8 > $OUT/run.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $OUT/kt_results.db > $OUT/stats.txt
python3 - <<PY >> $OUT/stats.txt
import sqlite3
c = sqlite3.connect("$OUT/kt_results.db")
rows = list(c.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
for name, b, e in rows[-14:]:
    print("%10.1f %8.1f  %s" % ((b - t0) / 1e3, (e - b) / 1e3, name[:60]))
PY
