"""Instruction-class census per basic block of one kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
python tools/isa_blocks.py file.s kernel_symbol [min_instructions]"""
import re
import sys
from collections import Counter

src, kern = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 50
txt = open(src).read()
i = txt.index('\n' + kern + ':')
j = txt.index('.end_amdhsa_kernel', i) if '.end_amdhsa_kernel' in txt[i:] else len(txt)
blocks = [['entry', []]]
for l in txt[i:j].split('\n'):
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append([m.group(1), []])
    elif l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        t = l.strip().split()
        if t:
            blocks[-1][1].append(t[0])


def cls(o):
    for pre, c in (('v_mfma', 'M'), ('global_load', 'L'), ('global_store', 'S'), ('v_accvgpr', 'a'), ('s_waitcnt', 'W'), ('s_barrier', 'B'),
                   ('scratch_load', 'x'), ('scratch_store', 'X'), ('ds_read', 'r'), ('ds_load', 'r'), ('ds_write', 'w'), ('ds_store', 'w'),
                   ('v_readlane', 'l'), ('v_writelane', 'l'), ('s_', 's'), ('v_', 'v')):
        if o.startswith(pre):
            return c
    return '?'


print('M mfma  v valu  a accvgpr-move  l read/writelane (SGPR spill)  x/X scratch load/store  r/w LDS  L/S global  s scalar  W waitcnt  B barrier')
for name, ins in blocks:
    if len(ins) >= minlen:
        print(name, len(ins), dict(Counter(cls(o) for o in ins)))
