#!/usr/bin/env python3
"""Static instruction budget of one kernel from hipcc -S output: instruction classes per source-marker section.

usage: isa_budget.py file.s <kernel-substring> [--dump]
Sections are split at `; SDPHASE <name>` comment lines emitted by asm volatile markers (if any); loops are not
weighted (static counts), so use together with the SQ counters.
"""
import re, sys, collections

def classify(op):
    if op.startswith('v_'):
        if op.startswith(('v_min_f64', 'v_max_f64')): return 'valu_minmax64'
        if '_f64' in op: return 'valu_f64'
        if op.startswith('v_cmp'): return 'valu_cmp'
        if op.startswith('v_cndmask'): return 'valu_cndmask'
        if op.startswith(('v_mov', 'v_accvgpr', 'v_readlane', 'v_readfirstlane', 'v_writelane')): return 'valu_mov'
        return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): return 'vmem_' + op.split('_')[0]
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'smem'
    if op.startswith('s_'): return 'salu'
    return 'other'

def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = '--dump' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[A-Za-z_]\S*:', l) and key in l.split(':')[0]:
            start = i
            break
    if start is None:
        sys.exit('kernel not found')
    sec = 'entry'
    counts = collections.OrderedDict()
    for l in lines[start + 1:]:
        if l.startswith('\t.end_amdhsa_kernel') or l.startswith('.Lfunc_end'):
            break
        s = l.strip()
        m = re.match(r';+\s*SDPHASE\s+(\S+)', s)
        if m:
            sec = m.group(1)
            continue
        if not s or s.startswith((';', '.')) or s.endswith(':'):
            continue
        op = s.split()[0]
        c = counts.setdefault(sec, collections.Counter())
        c[classify(op)] += 1
        c['_total'] += 1
        if dump:
            print(sec, s)
    classes = sorted({k for c in counts.values() for k in c})
    print('%-14s' % 'section' + ''.join('%15s' % k for k in classes))
    tot = collections.Counter()
    for sec, c in counts.items():
        print('%-14s' % sec + ''.join('%15d' % c[k] for k in classes))
        tot.update(c)
    print('%-14s' % 'TOTAL' + ''.join('%15d' % tot[k] for k in classes))

main()
