#!/usr/bin/env python3
"""PMC pass output (tools/pmc_profile.sh) -> an entry of profiles/pmc_latest.json: HBM bytes of the
perspective-write kernel PER PERSPECTIVE, corrected as MI355X_MICROARCH.md prescribes (WRITE_SIZE
exact for 16-byte streaming stores; FETCH_SIZE doubled; both counters in KiB).  bench.py multiplies
by its own perspectives per launch for `roofline.traffic` (same shape and dtype only).

usage: make_pmc_latest.py <pmc outdir> <profiles/pmc_latest.json>
The per-launch perspective count of the profiled run comes from the bench line that run printed
(<outdir>/wr.log): the RNG is counter-based and the population is in its steady state, so every run of
the same command sees the same perspectives per launch."""
import csv
import glob
import json
import os
import sys

root, outp = sys.argv[1], sys.argv[2]
line = [ln for ln in open(os.path.join(root, "wr.log")) if ln.startswith('{"metric"')][-1]
b = json.loads(line)
cfg = b["config"]
p_launch = b["perspectives_per_lattice"] * cfg["envs_per_gpu"]
tot = {"WRITE_SIZE": [0.0, 0], "FETCH_SIZE": [0.0, 0]}
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_persp_stream" in r["Kernel_Name"] and r["Counter_Name"] in tot:
            tot[r["Counter_Name"]][0] += float(r["Counter_Value"])
            tot[r["Counter_Name"]][1] += 1
w = tot["WRITE_SIZE"][0] / max(1, tot["WRITE_SIZE"][1]) * 1024
f = tot["FETCH_SIZE"][0] / max(1, tot["FETCH_SIZE"][1]) * 1024 * 2
esize = {"f32": 4, "f16": 2, "bf16": 2, "u8": 1}[cfg["out_dtype"]]
nq = 2 * cfg["d"] ** 2
alg = p_launch * (nq * esize + 12) + cfg["envs_per_gpu"] * nq
entry = {"d": cfg["d"], "out_dtype": cfg["out_dtype"], "envs": cfg["envs_per_gpu"], "p_error": cfg["p_error"],
         "perspectives_per_launch": p_launch, "hbm_bytes_per_launch": w + f, "write_bytes": w, "fetch_bytes_x2": f,
         "hbm_bytes_per_perspective": (w + f) / p_launch, "algorithmic_bytes_per_launch": alg,
         "traffic_over_algorithmic": (w + f) / alg, "dispatches": tot["WRITE_SIZE"][1],
         "source": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes, tools/pmc_profile.sh), "
                   "KiB units, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)"}
doc = {"entries": []}
if os.path.exists(outp):
    try:
        old = json.load(open(outp))
        doc["entries"] = [e for e in old.get("entries", []) if (e["d"], e["out_dtype"]) != (entry["d"], entry["out_dtype"])]
    except Exception:
        pass
doc["entries"].append(entry)
doc["entries"].sort(key=lambda e: (e["d"], e["out_dtype"]))
json.dump(doc, open(outp, "w"), indent=1)
print(json.dumps(entry, indent=1))
