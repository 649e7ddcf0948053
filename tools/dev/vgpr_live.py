#!/usr/bin/env python3
"""Live vector registers along one kernel of a `hipcc -S` listing (gfx9 syntax).

usage: vgpr_live.py file.s <kernel-substring> [--trace]

A backward data-flow pass over the kernel's basic blocks; prints, per `; SDPHASE <name>` section (asm volatile
markers, `-DSD_MARK` builds), the largest number of simultaneously live VGPRs and the instruction where it occurs.
Compile the kernel with a relaxed register budget (no spills) to see what each phase really needs.  Writes under a
partial exec mask are treated as full definitions (optimistic inside divergent regions).
"""
import re
import sys

NO_DEST = ('ds_write', 'ds_store', 'global_store', 'flat_store', 'buffer_store', 'scratch_store', 's_', 'v_cmp', 'v_cmpx',
           'global_atomic', 'ds_add_u32', 'ds_or', 'v_nop', 'buffer_wbl2', 'buffer_inv', 'ds_nop')
RETURNING = ('_rtn', 'global_atomic')  # atomics with sc0/glc return a value: handled by operand count heuristics


def regs_of(tok):
    tok = tok.strip()
    m = re.match(r'^v(\d+)$', tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def parse(lines):
    ins = []  # (label or None, op, defs, uses, text, phase)
    phase = 'entry'
    for l in lines:
        s = l.strip()
        m = re.match(r';+\s*SDPHASE\s+(\S+)', s)
        if m:
            phase = m.group(1)
            continue
        if not s or s.startswith((';', '.')):
            continue
        if s.endswith(':') and ' ' not in s:
            ins.append((s[:-1], None, [], [], s, phase))
            continue
        s = s.split(';')[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in re.split(r',(?![^\[]*\])', parts[1])] if len(parts) > 1 else []
        # strip modifiers glued to the last operand ("v1 offset:8")
        ops = [o.split()[0] if o else o for o in ops]
        defs, uses = [], []
        if op.startswith(NO_DEST) and '_rtn' not in op:
            for o in ops:
                uses += regs_of(o)
            if op.startswith('v_cmp') and ops and regs_of(ops[0]):
                pass
        else:
            if ops:
                defs += regs_of(ops[0])
            start = 1
            if op.startswith(('ds_read2', 'ds_load2')):
                start = 1
            for o in ops[start:]:
                uses += regs_of(o)
            # accumulating forms read their destination
            if op.startswith(('v_fmac', 'v_mac', 'v_dot')) or 'dpp' in s or 'sdwa' in s:
                uses += defs
            if op.startswith('v_mov_b32') and ('quad_perm' in s or 'row_' in s):
                uses += defs
        ins.append((None, op, defs, uses, s, phase))
    return ins


def main():
    path, key = sys.argv[1], sys.argv[2]
    trace = '--trace' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[A-Za-z_]\S*:', l) and key in l.split(':')[0]:
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    body = []
    for l in lines[start + 1:]:
        if l.startswith('\t.end_amdhsa_kernel') or l.startswith('.Lfunc_end') or l.strip().startswith('.section'):
            break
        body.append(l)
    ins = parse(body)
    n = len(ins)
    label_at = {t[0]: i for i, t in enumerate(ins) if t[0] is not None}
    succ = [[] for _ in range(n)]
    for i, (lab, op, d, u, s, ph) in enumerate(ins):
        if op is None:
            if i + 1 < n:
                succ[i].append(i + 1)
            continue
        if op == 's_endpgm':
            continue
        if op == 's_branch':
            tgt = s.split()[1]
            if tgt in label_at:
                succ[i].append(label_at[tgt])
            continue
        if op.startswith('s_cbranch'):
            tgt = s.split()[1]
            if tgt in label_at:
                succ[i].append(label_at[tgt])
        if i + 1 < n:
            succ[i].append(i + 1)
    live_in = [0] * n
    changed = True
    it = 0
    while changed:
        changed = False
        it += 1
        for i in range(n - 1, -1, -1):
            out = 0
            for j in succ[i]:
                out |= live_in[j]
            d = 0
            for r in ins[i][2]:
                d |= 1 << r
            u = 0
            for r in ins[i][3]:
                u |= 1 << r
            new = (out & ~d) | u
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    best = {}
    order = []
    for i in range(n):
        ph = ins[i][5]
        c = bin(live_in[i]).count('1')
        if ph not in best:
            best[ph] = (c, i)
            order.append(ph)
        elif c > best[ph][0]:
            best[ph] = (c, i)
        if trace and ins[i][1] is not None:
            print('%4d %-12s %s' % (c, ph, ins[i][4]))
    print('%-16s %6s  %s' % ('phase', 'live', 'at'))
    for ph in order:
        c, i = best[ph]
        print('%-16s %6d  %s' % (ph, c, ins[i][4][:90]))


if __name__ == '__main__':
    main()
