#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace, optional PMC counter collection) per kernel AND per grid size.
The same SpMV kernel runs on every multigrid level, so the per-kernel average of `--stats` mixes fine and coarse launches;
grouping by grid size separates the fine-level launches that bench.py's `roofline` refers to.

usage: summarize.py <dir with *_kernel_trace.csv [and *_counter_collection.csv]> [out.md]
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    lines = []
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    groups = collections.defaultdict(list)
    for f in traces:
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            groups[(short(r["Kernel_Name"]), grid)].append(dur)
    tot = sum(sum(v) for v in groups.values())
    lines.append("| kernel | grid (threads) | calls | total us | avg us | min us | max us | % |")
    lines.append("|---|---|---|---|---|---|---|---|")
    for (k, g), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %d | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k, g, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size", 0) or 0)
            pmc[(short(r["Kernel_Name"]), grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {}
    if pmc:
        lines.append("")
        lines.append("| kernel | grid | counter | launches | avg per launch |")
        lines.append("|---|---|---|---|---|")
        for (k, g), cs in sorted(pmc.items(), key=lambda kv: kv[0]):
            for c, v in sorted(cs.items()):
                lines.append("| `%s` | %d | %s | %d | %.6g |" % (k, g, c, len(v), sum(v) / len(v)))
                summary.setdefault("%s@%d" % (k, g), {})[c] = sum(v) / len(v)
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")
        if summary:
            json.dump(summary, open(os.path.splitext(out)[0] + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
