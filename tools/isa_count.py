"""Instruction mix of one kernel in a hipcc -S listing: python isa_count.py file.s <mangled-name-substring>."""
import re, sys, collections
s = open(sys.argv[1]).read()
for fn in sys.argv[2:]:
    starts = [m for m in re.finditer(r'^(_Z\w*%s\w*):' % re.escape(fn), s, re.M)]
    for m in starts:
        end = s.index('s_endpgm', m.end())
        body = s[m.end():end]
        ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = collections.Counter(ins)
        print(m.group(1)[:60], 'instrs', len(ins), 'valu', sum(v for k, v in c.items() if k.startswith('v_')),
              'pk', sum(v for k, v in c.items() if k.startswith('v_pk')), 'mul32', c['v_mul_lo_u32'] + c['v_mul_hi_u32'],
              'trans', sum(v for k, v in c.items() if k in ('v_exp_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_log_f32', 'v_sqrt_f32')))
        print('   ', c.most_common(22))
