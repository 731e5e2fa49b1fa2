#!/usr/bin/env python
"""Instruction histogram of the K loop(s) of a kernel in a hipcc -S dump:  isa_stats.py file.s <kernel-substring>
Prints per loop (a label with a backward branch to it) the MFMA / VALU / LDS / VMEM / SALU counts."""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(('E:', ':')) is not None and l.split(':')[0].find(key) >= 0)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(f'{body[0][:100]}  ({end - start} lines)')
    for a, b in loops:
        ops = Counter()
        for l in body[a:b + 1]:
            t = l.strip().split()
            if not t or t[0].startswith((';', '.')) or t[0].endswith(':'):
                continue
            op = t[0]
            if op.startswith('v_mfma'):
                ops['MFMA'] += 1
            elif op.startswith('v_'):
                ops['VALU'] += 1
                ops['  ' + op] += 1
            elif op.startswith('ds_'):
                ops['LDS'] += 1
                ops['  ' + op] += 1
            elif op.startswith(('buffer_', 'global_', 'flat_')):
                ops['VMEM'] += 1
            elif op.startswith('s_waitcnt'):
                ops['waitcnt'] += 1
            elif op.startswith('s_nop'):
                ops['nop'] += 1
            elif op.startswith('s_barrier'):
                ops['barrier'] += 1
            elif op.startswith('s_'):
                ops['SALU'] += 1
        if ops['MFMA'] == 0:
            continue
        print(f'  loop lines {a}-{b}: ' + ', '.join(f'{k} {ops[k]}' for k in ('MFMA', 'VALU', 'LDS', 'VMEM', 'SALU', 'waitcnt', 'nop', 'barrier')))
        print('     ' + ', '.join(f'{k.strip()} {v}' for k, v in sorted(ops.items(), key=lambda kv: -kv[1]) if k.startswith('  ')))


if __name__ == '__main__':
    main()
