#!/bin/bash
# The rows of DESIGN.md section 7 on one box: tools/table_run.sh <out dir>
OUT=${1:-gpurun_out/table}
mkdir -p "$OUT"
Q="--steps 60 --warmup 10 --nn-steps 0 --cpu-seconds 0 --no-shard-leg"
run() { name=$1; shift; timeout -k 10 300 python bench.py $Q "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "$name failed"; echo "$name done"; }
run d7_f32
run d7_torch_empty --stack-candidates 1 --stack-kinds torch
run d9_f32 --size 9 --p-error 0.15
run d9_chunks16 --size 9 --p-error 0.15 --chunks 16
run d11 --size 11 --envs 32768
run d13 --size 13 --envs 16384
run d15 --size 15 --p-error 0.08 --envs 16384
run d7_bf16 --out-dtype bf16
run d7_u8 --out-dtype u8
run d5 --size 5
run d3 --size 3
run d17 --size 17 --p-error 0.08 --envs 8192
run d19 --size 19 --p-error 0.07 --envs 8192
run d21 --size 21 --p-error 0.06 --envs 8192
python3 tools/table_summary.py "$OUT" | tee "$OUT/summary.txt"
