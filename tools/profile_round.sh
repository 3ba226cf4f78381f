#!/bin/bash
# One profiled configuration, on the GPU box (through gpurun):
#   tools/profile_round.sh <tag> [bench args...]
# -> gpurun_out/<tag>/bench.json            plain `python3 bench.py <args>` line
#    gpurun_out/<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the SAME command
#    gpurun_out/<tag>/kernel_trace_timed_region.txt   the stack write's dispatches of the timed region alone (tools/trace_tail.py)
#    gpurun_out/<tag>/bench_under_rocprof.json   the line that very run printed
#    gpurun_out/<tag>/pmc/summary.txt       separate --pmc passes (tools/pmc_profile.sh)
#    gpurun_out/pmc_latest.json             profiles/pmc_latest.json with the entry of this (d, dtype) updated
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
echo "[$TAG] plain bench"
python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[$TAG] kernel trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" "$@" --cpu-seconds 0 --nn-steps 0 --no-shard-leg \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
STEPS=$(echo "$@" | sed -n 's/.*--steps \([0-9]*\).*/\1/p'); STEPS=${STEPS:-200}
python3 "$REPO/tools/trace_tail.py" "$OUT/trace" k_persp_stream "$STEPS" > "$OUT/kernel_trace_timed_region.txt" 2>&1 || true
rm -rf "$OUT/trace"
cd "$REPO"
echo "[$TAG] pmc passes"
bash tools/pmc_profile.sh "$OUT/pmc" "$@" > /dev/null
[ -f "$REPO/gpurun_out/pmc_latest.json" ] || cp "$REPO/profiles/pmc_latest.json" "$REPO/gpurun_out/pmc_latest.json"
python3 tools/make_pmc_latest.py "$OUT/pmc" "$REPO/gpurun_out/pmc_latest.json" > "$OUT/pmc_entry.json"
find "$OUT/pmc" -name '*.csv' -size +1M -delete
echo "profiled $TAG"
