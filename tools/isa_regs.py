"""Register / spill table of the kernels in a hipcc device assembly file:  python tools/isa_regs.py /tmp/k.s [name substring]"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(_Z\w+)\n(.*?)\.vgpr_spill_count:\s+(\d+)', s, re.S):
    if len(sys.argv) > 2 and sys.argv[2] not in m.group(1):
        continue
    blk = m.group(2)
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, None])[1]
    print(m.group(1)[:90], 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'sgpr', g('sgpr_count'), 'scratch', g('private_segment_fixed_size'), 'spill', m.group(3))
