#!/bin/bash
# development: HBM write / fetch counters of the bench's kernels, one pass each: tools/pmc_write_pass.sh tag
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcw_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c -d $OUT/$c -o pmc -- python3 $R/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $OUT/$c.err || exit 1
  python3 $R/tools/pmc_dump.py $OUT/$c/pmc_results.db 1 >> $OUT/summary.txt
done
cat $OUT/summary.txt
