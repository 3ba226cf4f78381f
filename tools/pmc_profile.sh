#!/bin/bash
# PMC passes over a short bench run (run on the GPU box through gpurun).  Counters are collected
# in their own passes with no tracing, as MI355X_MICROARCH.md prescribes.
# usage: tools/pmc_profile.sh <outdir> [bench args...]
set -e
OUT=$(realpath -m "$1"); shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-seconds 0 --nn-steps 0 --no-shard-leg --no-events $*"
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pass sq3 SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass wr WRITE_SIZE GRBM_GUI_ACTIVE
pass rd FETCH_SIZE
python3 "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1 || true
cat "$OUT/summary.txt"
