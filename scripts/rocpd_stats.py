"""Per-kernel summary (count, total, average, share) from a rocprofv3 rocpd SQLite database."""
import re
import sqlite3
import sys


def main(path, top=45):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute("select s.%s, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id" % name_col).fetchall()
    agg = {}
    for name, dt in rows:
        n = re.sub(r'vaenpvc::(tuned|generic)::', '', name)
        n = re.sub(r'\(.*$', '', n)
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += dt
    tot = sum(v[1] for v in agg.values())
    print('%-110s %6s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-110s %6d %12.1f %10.1f %6.2f' % (n[:110], k, t / 1e3, t / 1e3 / k, 100.0 * t / tot))
    print('TOTAL kernel time us: %.1f over %d dispatches' % (tot / 1e3, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
