"""Per-kernel averages of the hardware counters in a rocprofv3 --pmc rocpd database.
usage: python tools/rocpd_pmc.py <results.db> [kernel-substring] [--each]      (--each: one line per dispatch, in launch order)"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    each = '--each' in sys.argv
    argv = [a for a in sys.argv if a != '--each']
    filt = argv[2] if len(argv) > 2 else ''
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda stem: [t for t in tabs if t.startswith(stem)][0]
    disp, sym, pmc, ev = T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol'), T('rocpd_info_pmc'), T('rocpd_pmc_event')
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    namecol = 'kernel_name' if 'kernel_name' in scols else 'display_name'
    q = ("select s.%s, d.id, d.end - d.start, p.name, sum(e.value) from %s e join %s p on e.pmc_id = p.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by d.id, p.name order by d.start" % (namecol, ev, pmc, disp, sym))
    if each:
        rows, order = defaultdict(dict), []
        for name, did, dt, cname, val in db.execute(q):
            if filt not in name:
                continue
            if did not in rows:
                order.append(did)
                rows[did]['name'], rows[did]['us'] = name, dt / 1e3
            rows[did][cname] = val
        import json
        for did in order:
            print(json.dumps(rows[did]))
        return
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    seen = set()
    for name, did, dt, cname, val in db.execute(q):
        if filt not in name:
            continue
        agg[name][cname].append(val)
        if did not in seen:
            seen.add(did)
            dur[name].append(dt)
    for name in agg:
        print("%s  launches=%d avg_us=%.2f" % (name[:90], len(dur[name]), sum(dur[name]) / len(dur[name]) / 1e3))
        for c, v in sorted(agg[name].items()):
            print("    %-32s avg %.4g   (min %.4g max %.4g)" % (c, sum(v) / len(v), min(v), max(v)))


if __name__ == '__main__':
    main()
