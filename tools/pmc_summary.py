#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for the GEMM kernels (averages per dispatch)."""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    f = max(glob.glob(d + '/*/*_counter_collection.csv'), key=os.path.getmtime)
    t = max(glob.glob(d + '/*/*_kernel_trace.csv'), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t)):
        dur[r['Kernel_Name'][:48]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for name in agg:
        if 'dense' in name or 'wgrad_kernel' in name or 'wgrad2_kernel' in name:
            c = {k: sum(v) / len(v) for k, v in agg[name].items()}
            us = sum(dur[name]) / len(dur[name])
            line = "%-50s %8.1f us " % (name, us)
            if 'GRBM_GUI_ACTIVE' in c:
                cyc = c['GRBM_GUI_ACTIVE'] / 8
                simd = cyc * 1024
                line += "clk %.2f GHz | mfma %.1f%% | waves/SIMD %.2f | of wave time: wait_any %.1f%% wait_inst %.1f%% active %.1f%%" % (
                    cyc / us / 1e3, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd, 4 * c['SQ_WAVE_CYCLES'] / simd,
                    100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'],
                    100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'])
            else:
                line += " ".join("%s=%.3g" % (k.replace('SQ_', ''), v) for k, v in sorted(c.items()))
            print(line)
