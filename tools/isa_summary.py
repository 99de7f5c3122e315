#!/usr/bin/env python
"""Per-kernel digest of a gfx950 ISA listing (tools/cc_one.sh <name> -S writes /tmp/<name>.s): where the waits, stores, barriers,
matrix instructions and scratch accesses sit, and which basic blocks form loops -- enough to check by eye (and in tests/test_boundary.py
by rule) that a hand-pipelined kernel has no `s_waitcnt vmcnt(0)` and no scratch traffic inside its tile loop.

    python tools/isa_summary.py /tmp/pwsweep.s [substring of the kernel name] [--dump LO HI]
"""
import re
import sys


def kernels(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)\n\.Lfunc_end\d+:', text, flags=re.S | re.M):
        out[m.group(1)] = m.group(2).split('\n')
    return out


def loops(lines):
    """[(first line, last line)] of backward branches: label position .. branch position"""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    res = []
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                res.append((labels[tgt], i))
    return res


PATS = ['scratch_', 's_waitcnt vmcnt', 'global_store', 'global_load', 'buffer_', 's_barrier', 'v_mfma', 'ds_read_b64_tr', 'ds_read_b128',
        'ds_write_b128', 'ds_write_b64', 'ds_read_b64 ', 'v_accvgpr']


def digest(lines):
    d = {}
    for pat in PATS:
        d[pat] = [i for i, l in enumerate(lines) if pat in l and not l.lstrip().startswith(';')]
    return d


if __name__ == '__main__':
    path = sys.argv[1]
    sel = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
    ks = kernels(path)
    for name, lines in ks.items():
        if sel not in name:
            continue
        print(name, '(%d lines)' % len(lines))
        lp = loops(lines)
        print('  loops (label line .. back branch):', lp)
        big = max(lp, key=lambda ab: ab[1] - ab[0]) if lp else None
        for pat, idx in digest(lines).items():
            if idx:
                inl = [i for i in idx if big and big[0] <= i <= big[1]]
                print('  %-16s %4d total, %4d in the longest loop  %s' % (pat, len(idx), len(inl), idx[:24]))
        if big:
            for i in range(big[0], big[1] + 1):
                if 's_waitcnt' in lines[i] and 'vmcnt' in lines[i]:
                    print('    loop wait @%d: %s' % (i, lines[i].strip()))
        if '--dump' in sys.argv:
            a = sys.argv.index('--dump')
            lo, hi = int(sys.argv[a + 1]), int(sys.argv[a + 2])
            for i in range(lo, hi):
                print('%5d %s' % (i, lines[i]))
