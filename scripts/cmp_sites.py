"""Side-by-side table of scripts/site_times.py outputs.   python scripts/cmp_sites.py a.txt b.txt ..."""
import sys
def rd(f):
    d = {}
    for l in open(f):
        p = l.split()
        if len(p) >= 3 and p[2] == 'us' and p[0] not in d:
            d[p[0]] = float(p[1])
    return d
tabs = [rd(f) for f in sys.argv[1:]]
keys = []
for t in tabs:
    for k in t:
        if k not in keys:
            keys.append(k)
print('%-14s' % 'site' + ''.join('%10s' % f.split('/')[-1][:9] for f in sys.argv[1:]))
for k in keys:
    print('%-14s' % k + ''.join('%10.1f' % t.get(k, 0) for t in tabs))
print('%-14s' % 'sum' + ''.join('%10.1f' % sum(t.values()) for t in tabs))
