#!/bin/bash
# Profiles one bench.py workload on the GPU box: a kernel trace (durations) and separate
# counter passes (one rocprofv3 run each; counters are never combined with tracing).
#   scripts/profile_gpu.sh <workload> <outdir under gpurun_out/> [extra bench args]
# Every step runs under its own time limit and the script stops at the first failure.
set -e
WL=${1:-coupled_vel}
OUT=${2:-prof_$WL}
shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/$OUT
rm -rf "$D" # (an earlier run's files, named by process id, would sit beside this run's)
mkdir -p "$D"
export TMPDIR=/tmp
cd "$ROOT"
BENCH="python3 bench.py --workload $WL --steps 5 --warmup 2 --plain $*"

timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- $BENCH > "$D/trace.log" 2>&1
echo "trace done"

pass() { # name, counters...
    local name=$1
    shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$D/pmc_$name" -- $BENCH > "$D/pmc_$name.log" 2>&1
    echo "pmc $name done"
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
pass busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
pass f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES
echo "all passes done"
