#!/usr/bin/env python3
"""dev: static instruction census of ONE device kernel by source line.

    hipcc -O3 ... -gline-tables-only -S --cuda-device-only rc_correct.hip -o x.s
    tools/asm_census.py x.s '<mangled kernel name or a substring of it>' [--by line|range] [--ranges FILE]

Every instruction of the kernel is attributed to the innermost `.loc` in front of it (file, line) and put in
a class: valu (v_*), salu (s_* without waits / branches), lds (ds_*), vmem (global_* / buffer_* / flat_* /
scratch_*), smem (s_load* / s_buffer_load*), branch, wait.  `--ranges FILE` folds lines into named source
ranges: one `name file first last` per line of FILE.  Straight-line code executes once per wave, so for the
threshold rows of the fused kernel static = dynamic; loops have to be weighted by hand (the tool prints the
basic-block label an instruction sits under with --blocks).
"""
import argparse
import collections
import re
import sys


def classify(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith(('s_load', 's_buffer_load', 's_store')):
        return 'smem'
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')):
        return 'wait'
    if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm')
    ap.add_argument('kernel')
    ap.add_argument('--ranges')
    ap.add_argument('--blocks', action='store_true')
    ap.add_argument('--top', type=int, default=40)
    a = ap.parse_args()
    files = {}
    lines = open(a.asm).read().split('\n')
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        if start is None and re.match(r'^(\S+):', ln) and a.kernel in ln.split(':')[0] and not ln.startswith('.'):
            start = i
    if start is None:
        sys.exit('kernel not found')
    ranges = []
    if a.ranges:
        for ln in open(a.ranges):
            p = ln.split()
            if len(p) == 4 and not ln.startswith('#'):
                ranges.append((p[0], p[1], int(p[2]), int(p[3])))
    by = collections.defaultdict(collections.Counter)
    blocks = collections.defaultdict(collections.Counter)
    cur = ('?', 0)
    blk = 'entry'
    tot = collections.Counter()
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith('.Lfunc_end') or s.startswith('.section') and 'rodata' in s:
            break
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r'^(\.LBB\S+):', s)
        if m:
            blk = m.group(1)
            continue
        if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        c = classify(op)
        if c == 'other':
            continue
        key = cur
        if ranges:
            key = ('(other)', 0)
            for name, f, lo, hi in ranges:
                if cur[0] == f and lo <= cur[1] <= hi:
                    key = (name, 0)
                    break
        by[key][c] += 1
        blocks[blk][c] += 1
        tot[c] += 1
    cls = ['valu', 'salu', 'lds', 'vmem', 'smem', 'branch', 'wait']
    print('%-44s' % 'where' + ''.join('%8s' % c for c in cls))
    print('%-44s' % 'TOTAL' + ''.join('%8d' % tot[c] for c in cls))
    order = sorted(by.items(), key=lambda kv: -kv[1]['valu'])
    if ranges:
        order = [(k, by[k]) for k in [(r[0], 0) for r in ranges] + [('(other)', 0)] if k in by]
    for (f, l), cnt in order[:a.top if not ranges else None]:
        name = f if ranges else '%s:%d' % (f, l)
        print('%-44s' % name + ''.join('%8d' % cnt[c] for c in cls))
    if a.blocks:
        print('\nby basic block (in order of appearance)')
        for b, cnt in blocks.items():
            print('%-44s' % b + ''.join('%8d' % cnt[c] for c in cls))


if __name__ == '__main__':
    main()
