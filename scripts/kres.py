"""Per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a compile).
usage: python scripts/kres.py remarks.txt [substring ...]   (demangles with c++filt)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, n in zip(blocks, dem):
    n = re.sub(r'vaenpvc::(tuned|generic)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    if len(sys.argv) > 2 and not any(k in n for k in sys.argv[2:]):
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    print('%-72s V=%3d A=%3d scratch=%4d occ=%d lds=%6d S=%3d' % (n[:72], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
                                                                 g(r'LDS Size \[bytes/block\]'), g('SGPRs')))
