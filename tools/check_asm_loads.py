#!/usr/bin/env python3
"""ISA safety check for hand-issued (inline-asm) global loads with VGPR destinations.

hipcc treats the destination of an asm load as written when the asm statement ends, so it is free
to copy / spill / reuse it before the data has landed.  This script walks the .s of the named
kernels along their control-flow graph (basic blocks, fixed point over the loops) and reports any
instruction that reads or writes a register that MAY be the destination of an asm `global_load_*` for
which no `s_waitcnt vmcnt(N)` with N small enough lies on some path to it ("in flight"), other than
the asm loads themselves.  The count model is the hardware's: loads retire in issue order, LDS-DMA
pieces count as well.

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


def tail_union(a, b):
    """May-be-in-flight merge of two queues of outstanding loads (oldest first), aligned at the YOUNGEST end: what
    `s_waitcnt vmcnt(N)` retires depends on how many younger loads there are."""
    if len(a) < len(b):
        a, b = b, a
    out = list(a)
    off = len(a) - len(b)
    for i, e in enumerate(b):
        out[off + i] = out[off + i] | e
    return tuple(out)


def parse_blocks(body):
    """[(label or None, [(line_no, text, in_asm)])], split at labels and after branches."""
    blocks = [[None, []]]
    in_asm = False
    for ln, line in enumerate(body.split('\n')):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        mm = re.match(r'(\.LBB\w+):', t)
        if mm:
            blocks.append([mm.group(1), []])
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        blocks[-1][1].append((ln, t, in_asm))
        if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm') or t.startswith('s_setpc'):
            blocks.append([None, []])
    return blocks


def check(src, name):
    m = re.search(re.escape(name) + r':(.*?)\.end_amdhsa_kernel', src, re.S)
    if not m:
        print("kernel not found:", name)
        return 1
    bad = 0
    # check 1 (linear walk): an SGPR written by a VALU instruction (v_readfirstlane / v_readlane) needs 5 wait states
    # before a vector-memory instruction reads it; the compiler's hazard recognizer does not see SGPR operands of asm
    in_asm = False
    valu_sgpr = {}      # sgpr -> instructions issued since the VALU write
    for ln, line in enumerate(m.group(1).split('\n')):
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
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

    # check 2 (dataflow over the control-flow graph): registers that MAY be the destination of an asm load still in
    # flight.  State = queue of outstanding asm VMEM ops, oldest first; merged at joins with tail_union; fixed point.
    blocks = parse_blocks(m.group(1))
    index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
    succ = []
    for i, (lab, ins) in enumerate(blocks):
        out = []
        last = ins[-1][1] if ins else ''
        if last.startswith('s_branch') or last.startswith('s_cbranch'):
            tgt = last.split()[-1]
            if tgt in index:
                out.append(index[tgt])
        if not (last.startswith('s_branch') or last.startswith('s_endpgm') or last.startswith('s_setpc')) and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    state_in = {0: ()}
    work = [0]
    hits = {}
    while work:
        i = work.pop()
        queue = list(state_in[i])
        for ln, t, asm in blocks[i][1]:
            if asm and t.startswith('global_load_lds'):
                queue.append(frozenset())
                continue
            if asm and t.startswith('global_load'):
                queue.append(frozenset(regs(t.split(',')[0])))
                continue
            if t.startswith('s_waitcnt') and 'vmcnt' in t:
                n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                del queue[:max(0, len(queue) - n)]
                continue
            if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_endpgm'):
                continue
            if queue:
                hit = regs(t) & frozenset().union(*queue)
                if hit:
                    hits[ln] = (sorted(hit)[:6], t[:80])
        if len(queue) > 64:        # a loop that issues without waiting: keep the youngest, the state must stay finite
            del queue[:len(queue) - 64]
        q = tuple(queue)
        for j in succ[i]:
            merged = q if j not in state_in else tail_union(state_in[j], q)
            if j not in state_in or merged != state_in[j]:
                state_in[j] = merged
                if j not in work:
                    work.append(j)
    for ln in sorted(hits)[:10]:
        print("  %s: line %d touches in-flight v%s: %s" % (name[:50], ln, hits[ln][0], hits[ln][1]))
    bad += len(hits)
    print("%-70s %s" % (name[:70], "OK" if not bad else "%d VIOLATIONS" % bad))
    return 1 if bad else 0


if __name__ == '__main__':
    s = open(sys.argv[1]).read()
    names = sys.argv[2:]
    if not names:
        names = sorted(set(re.findall(r'^(_ZN4svae\d+(?:dense_kernel|dense4_kernel|dense4_dual_kernel|wgrad_kernel|wgrad2_kernel|split_wgrad_kernel|dense_split\w*_kernel)\w+):', s, re.M)))
    sys.exit(max(check(s, n) for n in names))
