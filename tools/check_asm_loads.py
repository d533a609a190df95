#!/usr/bin/env python3
"""ISA safety check for hand-issued (inline-asm) global loads with VGPR destinations.

hipcc treats the destination of an asm load as written when the asm statement ends, so it is free
to copy / spill / reuse it before the data has landed.  This script walks the .s of the named
kernels linearly and reports any instruction that reads or writes a register that is the destination
of an asm `global_load_*` for which no asm `s_waitcnt vmcnt(N)` with N small enough has been seen
yet ("in flight"), other than the asm loads themselves.  The count model is the hardware's: loads
retire in issue order, LDS-DMA pieces count as well.

    python tools/check_asm_loads.py file.s kernel_name [...]
Exit status 1 if a violation is found.
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
SREG = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b')


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(src, name):
    m = re.search(re.escape(name) + r':(.*?)\.end_amdhsa_kernel', src, re.S)
    if not m:
        print("kernel not found:", name)
        return 1
    in_asm = False
    queue = []          # outstanding VMEM ops issued from asm, oldest first: set of dest regs (may be empty)
    bad = 0
    # second check: an SGPR written by a VALU instruction (v_readfirstlane / v_readlane) needs 5 wait states before a
    # vector-memory instruction reads it; the compiler's hazard recognizer does not see SGPR operands of inline asm
    valu_sgpr = {}      # sgpr -> instructions issued since the VALU write
    for ln, line in enumerate(m.group(1).split('\n')):
        t = line.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            nops = 1
            mm = re.match(r's_nop\s+(\d+)', t)
            if mm:
                nops = int(mm.group(1)) + 1
            for k in list(valu_sgpr):
                valu_sgpr[k] += nops
                if valu_sgpr[k] > 8:
                    del valu_sgpr[k]
            if in_asm and (t.startswith('global_load') or t.startswith('global_store')):
                stale = [r for r in sregs(t) if r in valu_sgpr and valu_sgpr[r] <= 5]
                if stale:
                    bad += 1
                    print("  %s: line %d reads VALU-written s%s too early: %s" % (name[:50], ln, stale, t[:80]))
            if t.startswith('v_readfirstlane') or t.startswith('v_readlane'):
                for r in sregs(t.split(',')[0]):
                    valu_sgpr[r] = 0
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        if in_asm and t.startswith('global_load_lds'):
            queue.append(set())
            continue
        if in_asm and t.startswith('global_load'):
            dst = t.split(',')[0]
            queue.append(regs(dst))
            continue
        if t.startswith('s_waitcnt') and 'vmcnt' in t:
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            while len(queue) > n:
                queue.pop(0)
            continue
        if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm'):
            continue
        inflight = set().union(*queue) if queue else set()
        hit = regs(t) & inflight
        if hit:
            bad += 1
            if bad <= 10:
                print("  %s: line %d touches in-flight v%s: %s" % (name[:50], ln, sorted(hit)[:6], t[:80]))
    print("%-70s %s" % (name[:70], "OK" if not bad else "%d VIOLATIONS" % bad))
    return 1 if bad else 0


if __name__ == '__main__':
    s = open(sys.argv[1]).read()
    names = sys.argv[2:]
    if not names:
        names = sorted(set(re.findall(r'^(_ZN4svae\d+(?:dense_kernel|dense4_kernel|wgrad_kernel|split_wgrad_kernel|dense_split\w*_kernel)\w+):', s, re.M)))
    sys.exit(max(check(s, n) for n in names))
