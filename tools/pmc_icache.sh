#!/bin/bash
# Instruction-cache and scalar-cache PMC passes over one scene (development aid). Usage: tools/pmc_icache.sh scene W H precision
set -o pipefail
S=${1:-dragons}; W=${2:-1920}; H=${3:-1080}; P=${4:-fp64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ic_$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$n -o pmc -- python3 $R/tools/one_scene.py $S $W $H $P 3 > /dev/null 2> $OUT/$n.err || exit 1
  python3 $R/tools/pmc_dump.py $OUT/$n/pmc_results.db 4 >> $OUT/summary.txt
}
: > $OUT/summary.txt
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ic2 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM
run ic3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY
cat $OUT/summary.txt
