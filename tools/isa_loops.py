"""Instruction counts of a kernel's loops from hipcc's device assembly (a quick VALU-bound estimate: a wave64 VALU instruction occupies a
16-lane SIMD for 4 cycles).   hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S -o /tmp/k.s file.hip
  python tools/isa_loops.py /tmp/k.s <substring of the mangled kernel name> [...]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', s, re.M)


def instrs(lines):
    out = []
    for ln in lines:
        t = ln.strip()
        if ln.startswith('\t') and t and not t.startswith(('.', ';')):
            out.append(t.split()[0])
    return out


for name in names:
    if not all(k in name for k in sys.argv[2:]):
        continue
    i = s.index('\n' + name + ':')
    j = s.index('.Lfunc_end', i)
    lines = s[i:j].split('\n')
    c = Counter(instrs(lines))
    print(name[:100], '\n  total', sum(c.values()), 'valu', sum(v for k, v in c.items() if k.startswith('v_')))
    labels = {}
    for n, ln in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            labels[m.group(1)] = n
    for n, ln in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            c2 = Counter(instrs(lines[labels[m.group(1)]:n]))
            print('  loop', m.group(1), 'instructions', sum(c2.values()), 'valu', sum(v for k, v in c2.items() if k.startswith('v_')),
                  'vmem', sum(v for k, v in c2.items() if k.startswith(('global_', 'buffer_'))), 'lds', sum(v for k, v in c2.items() if k.startswith('ds_')))
            print('     ', c2.most_common(16))
