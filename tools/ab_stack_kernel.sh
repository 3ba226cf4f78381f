#!/bin/bash
# A/B of the stack-write kernels in the real step loop on one box, one process each, alternating:
#   tools/ab_stack_kernel.sh <tag> [bench args]      VARIANTS="lattice stream:0 windows:0 ..." REPS=3
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --nn-steps 0 --no-shard-leg $@"
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-lattice stream:0 stream:2 windows:0 windows:1}; do
    k=${v%%:*}; c=${v##*:}; [ "$c" = "$v" ] && c=0
    TORIC_STACK_KERNEL=$k TORIC_STREAM_CFG=$c $B > $OUT/${k}${c}_$rep.json 2> $OUT/${k}${c}_$rep.err || echo "$v failed"
  done
done
python - <<PY
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        name=f.split("/")[-1].rsplit("_",1)[0]
        acc[name].append((j["value"]/1e6, r["achieved"], 1e3*r["avg_launch_ms"], 1e3*j["ms_per_step"]))
    except Exception as e:
        print(f, "unreadable", e)
for name,v in acc.items():
    print("%-12s" % name, "  ".join("%6.1f M %5.0f GB/s %6.1f us" % (a,b,c) for a,b,c,_ in v), "   step-launch %.1f us" % (sum(x[3]-x[2] for x in v)/len(v)))
PY
