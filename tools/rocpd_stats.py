"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database — the same table
`rocprofv3 --kernel-trace --stats` prints, usable when only the .db travelled back from the GPU box.
usage: python tools/rocpd_stats.py gpurun_out/prof/r1_results.db [--skip-first N] > profiles/xxx_kernel_stats.md"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    namecol = 'kernel_name' if 'kernel_name' in scols else ('display_name' if 'display_name' in scols else 'name')
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, disp, sym, namecol))
    rows = list(db.execute(q))
    total = float(sum(r[2] for r in rows))
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        print("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print("\ntotal kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main()
