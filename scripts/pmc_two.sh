#!/bin/bash
# Two counter passes (instruction mix, busy/wait cycles) of one bench.py workload.
#   scripts/pmc_two.sh <workload> <outdir under gpurun_out/>
set -e
WL=${1:-decoupled}
OUT=${2:-pmc_$WL}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/$OUT
mkdir -p "$D"
export TMPDIR=/tmp
cd "$ROOT"
BENCH="python3 bench.py --workload $WL --steps 3 --warmup 1 --cpu-sample 0 --host-steps 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d "$D/insts" -- $BENCH > "$D/insts.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$D/busy" -- $BENCH > "$D/busy.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$D/lds" -- $BENCH > "$D/lds.log" 2>&1
echo passes done
