#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace, optional PMC counter collection) per kernel AND per grid size.
The same SpMV kernel runs on every multigrid level, so the per-kernel average of `--stats` mixes fine and coarse launches;
grouping by grid size separates the fine-level launches that bench.py's `roofline` refers to.

If the run launched phase markers (fh_profile_marker -> kernels named k_phase_marker<N>; bench.py does), a second table splits every kernel's
launches by phase: bench.py's phases are 1-2 the timed steps (assembly + cycle), 3-4 cycles alone, 5-6 the fused sweep issued launch by launch,
7-8 the same launches replayed from one hipGraph, 9-10 one whole MGsolve (assembly + preparation + GMRES), 11-12 its GMRES alone; `steps` and `cycles`
rows of the fine-level `k_spmv_lx<2048, 3, ...>` are the IN-CYCLE launches bench.py's `roofline.frac` is quoted on.  A third table gives, per phase, the
span between its two markers, the sum of the kernel durations inside it and the idle share (launch gaps, host round trips).

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
    launches = []
    for f in traces:
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            groups[(short(r["Kernel_Name"]), grid)].append(dur)
            launches.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), grid, dur))
    tot = sum(sum(v) for v in groups.values())
    lines.append("| kernel | grid (threads) | calls | total us | avg us | min us | max us | % |")
    lines.append("|---|---|---|---|---|---|---|---|")
    for (k, g), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %d | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k, g, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    # ---- by phase (marker kernels) ----
    PHASES = {(1, 2): "steps (assembly + cycle)", (3, 4): "cycles", (5, 6): "sweep, launch by launch", (7, 8): "sweep, graph replay",
              (9, 10): "solve (assembly + preparation + GMRES)", (11, 12): "solve: GMRES alone"}
    marks = {}
    for t, k, g, dur in launches:
        if k.startswith("k_phase_marker<"):
            marks.setdefault(int(k[len("k_phase_marker<"):].rstrip(">")), []).append(t)
    if marks:
        byphase = collections.defaultdict(list)
        for t, k, g, dur in launches:
            if k.startswith("k_phase_marker<"):
                continue
            for (a, b), name in PHASES.items():
                if a in marks and b in marks and any(ta < t < tb for ta, tb in zip(marks[a], marks[b])):
                    byphase[(name, k, g)].append(dur)
        lines.append("")
        lines.append("| phase | kernel | grid (threads) | calls | avg us | min us | max us |")
        lines.append("|---|---|---|---|---|---|---|")
        for (name, k, g), v in sorted(byphase.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
            if sum(v) >= 20.0:
                lines.append("| %s | `%s` | %d | %d | %.2f | %.2f | %.2f |" % (name, k, g, len(v), sum(v) / len(v), min(v), max(v)))
        # span of a phase (marker to marker) against the kernel time inside it
        lines.append("")
        lines.append("| phase | occurrences | span us (marker to marker) | kernel time us | idle % (launch gaps, host round trips, copies) |")
        lines.append("|---|---|---|---|---|")
        for (a, b), name in PHASES.items():
            if a in marks and b in marks:
                spans = [(ta, tb) for ta, tb in zip(marks[a], marks[b]) if tb > ta]
                if not spans:
                    continue
                span = sum(tb - ta for ta, tb in spans) / 1e3
                busy = sum(dur for t, k, g, dur in launches if not k.startswith("k_phase_marker<") and any(ta < t < tb for ta, tb in spans))
                lines.append("| %s | %d | %.1f | %.1f | %.1f |" % (name, len(spans), span, busy, 100.0 * max(span - busy, 0.0) / max(span, 1e-9)))
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
