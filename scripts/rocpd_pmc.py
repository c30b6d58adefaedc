"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (one row per kernel name)."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, top=40):
    c = sqlite3.connect(path)
    def cols(t):
        return [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    pe, ip = cols('rocpd_pmc_event'), cols('rocpd_info_pmc')
    ksym = cols('rocpd_info_kernel_symbol')
    name_col = 'display_name' if 'display_name' in ksym else 'kernel_name'
    # event_id links pmc_event -> kernel_dispatch.event_id
    q = ("select s.%s, p.name, e.value, d.end - d.start, d.id from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id" % name_col)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(dict)
    for name, pmc, val, dt, did in c.execute(q):
        n = re.sub(r'vaenpvc::(tuned|generic)::', '', name)
        n = re.sub(r'\(.*$', '', n)
        agg[n][pmc] += val
        cnt[n].add(did)
        dur[n][did] = dt
    pmcs = sorted({p for v in agg.values() for p in v})
    print('%-86s %5s %9s ' % ('kernel', 'calls', 'avg_us') + ' '.join('%14s' % p[-14:] for p in pmcs))
    rows = sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]].values()))[:top]
    for n, v in rows:
        k = len(cnt[n])
        print('%-86s %5d %9.1f ' % (n[:86], k, sum(dur[n].values()) / k / 1e3) + ' '.join('%14.4g' % (v[p] / k) for p in pmcs))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
