#!/bin/bash
# SQ counters per kernel of one scene on the GPU box: tools/pmc_scene.sh name W H precision tag
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SALU -d $OUT -o pmc -- python3 $GRAFT_REPO_ROOT/tools/one_scene.py $1 $2 $3 $4 4 > $OUT/run.log 2>&1 || exit 1
python3 - <<PY > $OUT/stats.txt
import sqlite3
c = sqlite3.connect("$OUT/pmc_results.db")
for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    print("%-40s %-22s n=%-3d avg=%.0f" % (k[:40], cn, n, avg))
PY
