#!/bin/bash
# What state is this box's GPU in while the stack write runs?  (run on the GPU box)
#   tools/box_state.sh <out dir>
# Starts tools/placement_bench (the product's stream kernel on one VMM buffer, several mappings: ~10 s of back-to-back
# launches) and samples rocm-smi clocks / power / temperatures five times meanwhile.
OUT=${1:-gpurun_out/box_state}
mkdir -p "$OUT"
rocm-smi --showclocks --showpower --showtemp --showperflevel > "$OUT/smi_idle.txt" 2>&1
tools/placement_bench 6 4 1 > "$OUT/place.txt" 2>&1 &
PB=$!
for i in 1 2 3 4 5; do
    sleep 1.5
    rocm-smi --showclocks --showpower --showtemp > "$OUT/smi_load_$i.txt" 2>&1
done
wait $PB
grep -h "creation order" "$OUT/place.txt" | head -3
grep -h "fclk\|mclk\|sclk\|Power (W)\|memory) (C)\|junction" "$OUT/smi_load_3.txt"
