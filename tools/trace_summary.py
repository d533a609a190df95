#!/usr/bin/env python3
"""Per-step kernel time from a rocprofv3 --kernel-trace CSV of bench.py: busy time, gaps, and the kernels grouped by family.

    python tools/trace_summary.py <dir with *_kernel_trace.csv> [--steps 20]
"""
import collections
import csv
import glob
import os
import re
import sys


def family(name):
    if "dense_kernel" in name or "dense4_" in name or "dense_split" in name:
        return "decoder GEMM fwd/dgrad"
    if "wgrad_kernel" in name or "wgrad2_kernel" in name or "split_wgrad" in name:
        return "decoder GEMM wgrad"
    if name.startswith("Cijk") or "Cijk_" in name:
        return "encoder GEMM (hipBLASLt)"
    if "svae::" in name:
        return "svae::" + re.sub(r"^.*svae::([a-z0-9_]+).*$", r"\1", name)
    if "at::native" in name:
        m = re.search(r"(tanh_backward|tanh_kernel|FillFunctor|MulFunctor|normal|copy|sum|add)", name)
        return "ATen " + (m.group(1) if m else "other")
    return name[:40]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    f = max(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the timed region = the last `steps` repetitions; a step ends with its (single) Adam launch
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    start_i = ends[-steps - 1] + 1 if len(ends) > steps else 0
    rows = rows[:ends[-1] + 1]
    sel = rows[start_i:]
    t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    fam = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        k = family(r["Kernel_Name"])
        fam[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        fam[k][1] += 1
    print("steps %d: wall %.4f ms/step, kernels busy %.4f ms/step, gaps %.4f ms/step, %d launches/step"
          % (steps, (t1 - t0) / steps / 1e6, busy / steps / 1e6, (t1 - t0 - busy) / steps / 1e6, len(sel) // steps))
    for k, (ns, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print("  %-44s %8.2f us/step  (%.1f launches/step, %.2f us each)" % (k, ns / steps / 1e3, cnt / steps, ns / cnt / 1e3))


if __name__ == "__main__":
    main()
