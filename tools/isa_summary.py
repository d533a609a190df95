#!/usr/bin/env python3
"""Compress a kernel's ISA into a one-line event string (loads, waits, MFMAs, barriers, spills)."""
import re
import sys


def summarize(s, name, maxlen=1500):
    m = re.search(re.escape(name) + r':(.*?)\.end_amdhsa_kernel', s, re.S)
    out = []
    for l in m.group(1).split('\n'):
        t = l.strip()
        tag = None
        if t.startswith('s_waitcnt'): tag = t.replace('s_waitcnt ', 'w:')
        elif t.startswith('global_load_lds'): tag = 'G'
        elif t.startswith('global_load'): tag = 'g'
        elif t.startswith('global_store'): tag = 'W'
        elif t.startswith('scratch_'): tag = 'S'
        elif t.startswith('ds_read'): tag = 'L'
        elif t.startswith('s_barrier'): tag = 'BAR'
        elif t.startswith('.LBB'): tag = '|'
        elif t.startswith('s_cbranch'): tag = 'br'
        elif t.startswith('v_mfma'): tag = 'M'
        elif t.startswith('v_accvgpr'): tag = 'a'
        if tag is None:
            continue
        if out and out[-1][0] == tag:
            out[-1][1] += 1
        else:
            out.append([tag, 1])
    print(name)
    print(' '.join(t if n == 1 else '%sx%d' % (t, n) for t, n in out)[:maxlen])
    print()


if __name__ == '__main__':
    src = open(sys.argv[1]).read()
    for nm in sys.argv[2:]:
        summarize(src, nm)
