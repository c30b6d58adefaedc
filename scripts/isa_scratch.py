"""Where a kernel touches scratch (spill stores / reloads) in a `hipcc -S --cuda-device-only` dump, with the wait in front of each access.
usage: python scripts/isa_scratch.py dump.s <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().endswith(':') or (l.startswith('_ZN') and key in l and ': ;' in l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
print(lines[start][:120], len(body), 'lines;', sum('v_mfma' in l for l in body), 'mfma;', sum('scratch_' in l for l in body), 'scratch ops;',
      sum('v_accvgpr' in l for l in body), 'accvgpr moves')
for i, l in enumerate(body):
    if 'scratch_' in l:
        print(i, ' | '.join(x.strip().split(';')[0] for x in body[max(0, i - 2):i + 2])[:200])
