"""Turn the FETCH_SIZE / WRITE_SIZE counter passes of tools/profile_round.sh into profiles/<tag>_traffic.json.

usage: python3 tools/traffic_json.py <fetch_dir> <write_dir> <out.json>
Both counters are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports
wide coalesced streaming reads by 2x, so corrected = (2*FETCH + WRITE) * 1024; for kernels whose reads are 4 B/lane
(the dense kernels' row operand) that correction is an upper bound and the raw figure a lower bound.
"""
import collections
import csv
import glob
import os
import json
import re
import sys


def per_kernel(d):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        if "svae" in name:
            agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    what = sys.argv[4] if len(sys.argv) > 4 else "BASELINE cfg 2"
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/kbench.py at " + what +
                   "; KB per dispatch. gfx950: FETCH_SIZE reads exactly half the bytes of wide (16 B/lane) "
                   "coalesced streaming reads (MI355X_MICROARCH.md, HBM): 'hbm_bytes_corrected' = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024; dense4_kernel and wgrad_kernel read with 16 B/lane (the correction applies); "
                   "dense_kernel's row operand is read with 4 B/lane loads, for which the counter is uncalibrated, so its "
                   "corrected figure is an upper bound.",
           "kernels": {}}
    for k in fetch:
        f, w = fetch[k], write.get(k, 0.0)
        out["kernels"][k] = {"fetch_size_kb": round(f, 1), "write_size_kb": round(w, 1),
                             "hbm_bytes_raw": int((f + w) * 1024), "hbm_bytes_corrected": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-70s raw %8.1f MB  corrected %8.1f MB" % (k[:70], v["hbm_bytes_raw"] / 1e6, v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
