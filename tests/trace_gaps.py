"""Reads a rocprofv3 --kernel-trace csv of `bench.py` and reports where the GPU idles inside a step: a step = from one launch of the
element-matrix kernel to the next; busy = sum of kernel durations, idle = span - busy, with the largest gaps named by the kernels around them.
usage: python tests/trace_gaps.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
f = max(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)      # (child processes of the command leave smaller traces)
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_elem_q2hex_mfma" in r[2] or "k_cluster_q2hex_sf" in r[2]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    span = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:] + [rows[b]]):
        gaps.append((s1 - e0, n0.split("(")[0][:50], n1.split("(")[0][:50]))
    gaps.sort(reverse=True)
    steps.append({"n_kernels": len(seg), "span_us": span / 1e3, "busy_us": busy / 1e3, "idle_us": (span - busy) / 1e3,
                  "top_gaps_us": [(g / 1e3, a_, b_) for g, a_, b_ in gaps[:6]]})
# the timed region of bench.py: consecutive steps with the same kernel count
import collections
cnt = collections.Counter(s["n_kernels"] for s in steps)
# the timed steps: the most frequent kernel count among steps with a whole cycle in them (assembly-only loops of the roofline measurement have 2-3 kernels)
common = collections.Counter(s["n_kernels"] for s in steps if s["n_kernels"] >= 20).most_common(1)[0][0]
timed = [s for s in steps if s["n_kernels"] == common]
out = {"trace": os.path.basename(f), "steps_seen": len(steps), "kernels_per_step": common,
       "median_span_us": sorted(s["span_us"] for s in timed)[len(timed) // 2], "median_busy_us": sorted(s["busy_us"] for s in timed)[len(timed) // 2],
       "median_idle_us": sorted(s["idle_us"] for s in timed)[len(timed) // 2], "example_step": timed[len(timed) // 2]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
