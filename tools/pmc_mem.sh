#!/bin/bash
# Memory-pipeline PMC passes over one scene (development aid). Usage: tools/pmc_mem.sh scene W H precision
set -o pipefail
S=${1:-dragons}; W=${2:-1920}; H=${3:-1080}; P=${4:-fp64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mem_$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$n -o pmc -- python3 $R/tools/one_scene.py $S $W $H $P 3 > /dev/null 2> $OUT/$n.err || exit 1
  python3 $R/tools/pmc_dump.py $OUT/$n/pmc_results.db 4 >> $OUT/summary.txt
}
: > $OUT/summary.txt
run sq1 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run tcp2 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
run tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TA_BUSY_sum
cat $OUT/summary.txt
