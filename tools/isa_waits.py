"""Condensed view of a kernel's memory operations and waits (from hipcc -S output): global / LDS-write / barrier / branch / s_waitcnt
lines in program order with the number of matrix instructions between them.  usage: python tools/isa_waits.py file.s <mangled substring>"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
st = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l) and name in l][0]
end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
out, mf = [], 0
for l in lines[st:end]:
    t = l.strip()
    if not t or t.startswith(';'):
        continue
    op = t.split()[0]
    if 'mfma' in op:
        mf += 1
        continue
    if op.startswith(('global_', 'buffer_', 's_waitcnt', 's_barrier', 's_cbranch', 's_branch', 'ds_write', 'scratch_')) or re.match(r'^\.LBB', t):
        if mf:
            out.append(f'   [{mf} mfma]')
            mf = 0
        out.append(t.split(';')[0][:100])
print('\n'.join(out))
