#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, then HBM PMC passes.
# Usage: tools/profile.sh <round tag>
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o kt -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/bench_under_rocprof.json 2> $REPO/$OUT/kt.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $REPO/$OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_fetch.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $REPO/$OUT/pmc_write -o pmc -- python3 $REPO/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_write.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $REPO/$OUT/pmc_sq -o pmc -- python3 $REPO/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_sq.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc TD_TD_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $REPO/$OUT/pmc_mem -o pmc -- python3 $REPO/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_mem.err || exit 1
timeout -s KILL 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $REPO/$OUT/pmc_cache -o pmc -- python3 $REPO/bench.py --steps 6 --warmup 4 --min-seconds 0 --no-cpu-baseline > /dev/null 2> $REPO/$OUT/pmc_cache.err || exit 1
ls $REPO/$OUT
