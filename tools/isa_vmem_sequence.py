#!/usr/bin/env python3
"""Order of the vector-memory instructions, the s_waitcnt vmcnt(..) and the MFMA runs of ONE kernel in a device assembly file
(hipcc --cuda-device-only -S): D = LDS-DMA, l = global load, s = global store, x = scratch (spill) access, [n] = s_waitcnt vmcnt(n),
Mn = n MFMAs, | = basic-block boundary.  The counted waits of the register kernels (riccati_backward_rv / _rw.hpp) rely on this
order; a spill or a compiler-inserted vmcnt(0) shows here before it costs GPU time.
usage: isa_vmem_sequence.py file.s mangled-kernel-name-prefix"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(name) and '@' in l][0]
end = [i for i in range(start, len(lines)) if '.Lfunc_end' in lines[i]][0]
seq = []
for l in lines[start:end]:
    t = l.strip()
    if 'global_load_lds' in t:
        seq.append('D')
    elif t.startswith('global_load'):
        seq.append('l')
    elif t.startswith('global_store'):
        seq.append('s')
    elif t.startswith('scratch_'):
        seq.append('x')
    elif 's_waitcnt' in t and 'vmcnt' in t:
        seq.append('[%s]' % t.split('vmcnt(')[1].split(')')[0])
    elif t.startswith('v_mfma'):
        if seq and seq[-1].startswith('M'):
            seq[-1] = 'M%d' % (int(seq[-1][1:]) + 1)
        else:
            seq.append('M1')
    elif t.startswith('.LBB') or t.startswith('s_cbranch'):
        seq.append('|')
out = ''.join(seq)
for ch in 'Dlsx':
    out = re.sub('(%s+)' % ch, lambda m: '%s%d ' % (ch, len(m.group(1))), out)
print(re.sub(r'\|+', '|', out))
