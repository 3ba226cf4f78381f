#!/usr/bin/env python3
"""One line per bench line of a tools/table_run.sh directory: table_summary.py <dir>"""
import glob, json, os, sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:                                   # a run that failed leaves no line
        print(f"{os.path.basename(f)[:-5]:<16} no line ({e})")
        continue
    r = b["roofline"]
    sh = r.get("workgroup_shares", {})
    pr = sh.get("probe", {})
    print(f"{os.path.basename(f)[:-5]:<16} {b['config'].get('envs_per_gpu', '?'):>7} lattices "
          f"{(b['value'] or 0) / 1e6:8.2f} M env-steps/s  ms/step {b['ms_per_step']:.4f}  write {r['avg_launch_ms']:.4f} ms = {r['frac']:.3f} of 8 TB/s  "
          f"non-write {1e3 * (b['ms_per_step'] - r['avg_launch_ms'] * r.get('launches_per_step', 1)):5.1f} us  timed/probe {r.get('timed_over_probe', 0):.3f}  "
          f"default buffer {r.get('default_buffer', {}).get('frac', 0):.3f}  xcd bias {sh.get('xcd_bias')}"
          + (f" (probe: {pr.get('write_ms_biased', 0):.4f} against {pr.get('write_ms_equal_shares', 0):.4f} ms with equal shares)" if "write_ms_biased" in pr else "")
          + f"  verified {b.get('stack_verified', {}).get('ok')}")
