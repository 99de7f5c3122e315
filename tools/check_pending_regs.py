#!/usr/bin/env python
"""Static check of the ISA of csrc/dwroll.hip and csrc/pwsweep.hip (no GPU needed): the row requests of the row-pipelined depthwise
kernels and the tile requests of the one-sweep 1x1 backward are inline
assembly (global_load into registers the compiler does not know to be pending), so NO compiler-generated instruction may read
a request's destination registers between the request and the hand-placed s_waitcnt that covers it.  The one way the compiler
does that on its own is a register copy hoisted in front of the (tied-operand) wait statement; this script compiles the file to
assembly and fails if any v_mov / v_accvgpr / scratch store outside the ASM blocks reads a request destination register."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'torch_semantic_segmentation_amd', 'csrc')


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


FILES = (('dwroll.hip', 'roll'), ('pwsweep.hip', 'pwsweep'))


def check(asm_text, key='roll'):
    problems = []
    for name, body in re.findall(r'^(_ZN[^\n:]*%s[^\n:]*):.*?\n(.*?)s_endpgm' % key, asm_text, flags=re.S | re.M):
        lines = body.split('\n')
        # up to the final drain (the last hand-placed s_waitcnt vmcnt(0)): behind it the registers are ordinary again
        last = max([i for i, ln in enumerate(lines) if 's_waitcnt vmcnt(0)' in ln and i > 0 and 'ASMSTART' in lines[i - 1]] or [len(lines)])
        lines = lines[:last]
        pending, in_asm, first_req = set(), False, None
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if t.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if in_asm and t.startswith('global_load_dword'):
                pending |= regs(t.split()[1].rstrip(','))
                if first_req is None:
                    first_req = i       # nothing is in flight before the first request (zero-initialisation of the registers sits there)
        # every asm request destination is "pending-capable" for the whole kernel: a compiler-made copy FROM one is suspicious
        in_asm = False
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(';;#ASMSTART'):
                in_asm = True
            elif t.startswith(';;#ASMEND'):
                in_asm = False
            elif not in_asm and first_req is not None and i > first_req and re.match(r'(v_mov_b32|v_mov_b64|v_pk_mov_b32|v_swap_b32|v_accvgpr_write|scratch_store|buffer_store)', t):
                ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
                srcs = set()
                for o in ops[1:]:
                    srcs |= regs(o)
                if srcs & pending:
                    problems.append('%s: line %d: %s' % (name, i, t))
    return problems


def main():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    problems = []
    with tempfile.TemporaryDirectory() as d:
        for src, key in FILES:
            out = os.path.join(d, src.replace('.hip', '.s'))
            res = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S',
                                  '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', os.path.join(CSRC, src), '-o', out],
                                 check=True, capture_output=True, text=True)
            problems += check(open(out).read(), key)
            # a spilled register may be one with a request in flight (seen: a memory fault in an experimental build with 100 bytes
            # of scratch): the hand-pipelined kernels must not spill at all
            name = None
            for ln in res.stderr.split('\n'):
                m = re.search(r'Function Name: (\S+)', ln)
                if m:
                    name = m.group(1)
                m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', ln)
                if m and name and key in name and int(m.group(1)) != 0:
                    problems.append('%s: %s bytes of scratch per lane' % (name, m.group(1)))
    for pr in problems:
        print(pr)
    print('%d compiler-made reads of request destination registers' % len(problems))
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
