"""Kernel resource table (VGPRs / AGPRs / scratch / spills / occupancy) from the compiler:
  hipcc ... -Rpass-analysis=kernel-resource-usage -c rtoc_capi.hip 2> usage.txt; python tools/resource_usage.py usage.txt"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: Function Name: ', txt)[1:]
names = [b.split()[0] for b in blocks]
if not names:
    sys.exit("no kernel-resource-usage remarks in " + sys.argv[1] + " (did the compile fail?)")
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')
print("# kernel, VGPRs, AGPRs, ScratchSize[bytes/lane], VGPRs Spill, SGPRs, SGPRs Spill, Occupancy[waves/SIMD], LDS[bytes/block]")
for b, n in zip(blocks, dem):
    g = lambda k: (re.search(re.escape(k) + r': (\S+)', b) or [None, '?'])[1]
    print(", ".join([n.replace('void ', ''), g('VGPRs'), g('AGPRs'), g('ScratchSize [bytes/lane]'), g('VGPRs Spill'), g('TotalSGPRs'),
                     g('SGPRs Spill'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')]))
