"""tools/rocpd_pmc.py text (per-kernel counter averages of the conv_probe passes) -> the JSON bench.py reads
(profiles/rNN_pmc_conv.json): HBM bytes per launch (FETCH_SIZE doubled: MI355X_MICROARCH.md, gfx950 reports half of wide
coalesced reads; WRITE_SIZE as reported) and matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).   usage: python tools/pmc_conv_summary.py <txt>"""
import json
import re
import sys


def main():
    kern, cur = {}, None
    for line in open(sys.argv[1]):
        m = re.match(r'(\S.*?)\s+launches=(\d+) avg_us=([\d.]+)', line)
        if m:
            name = re.sub(r'^_Z\d+', '', m.group(1))[:40]
            cur = kern.setdefault(m.group(1), {'launches': int(m.group(2)), 'avg_us': float(m.group(3)), 'counters': {}})
            continue
        m = re.match(r'\s+(\S+)\s+avg ([\d.e+-]+)', line)
        if m and cur is not None:
            cur['counters'][m.group(1)] = float(m.group(2))
    tot_l = sum(k['launches'] for k in kern.values())
    w = lambda f: sum(f(k) * k['launches'] for k in kern.values() if f(k) is not None) / max(1, sum(k['launches'] for k in kern.values() if f(k) is not None))
    c = lambda k, n: k['counters'].get(n)
    read_mb = w(lambda k: None if c(k, 'FETCH_SIZE') is None else 2.0 * c(k, 'FETCH_SIZE') * 1024 / 1e6)
    write_mb = w(lambda k: None if c(k, 'WRITE_SIZE') is None else c(k, 'WRITE_SIZE') * 1024 / 1e6)
    busy = w(lambda k: None if (c(k, 'SQ_VALU_MFMA_BUSY_CYCLES') is None or not c(k, 'GRBM_GUI_ACTIVE')) else
             c(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * c(k, 'GRBM_GUI_ACTIVE') / 8.0))
    hit = w(lambda k: None if c(k, 'TCC_HIT_sum') is None else c(k, 'TCC_HIT_sum') / max(1.0, c(k, 'TCC_HIT_sum') + c(k, 'TCC_MISS_sum')))
    out = {'source': 'rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE | LDS + TCC; --kernel-trace only) over tools/conv_probe.py: '
                     'the 10 forward / data-gradient convolution launches of one step of the headline configuration, 4 repetitions; '
                     'FETCH_SIZE doubled (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE uncalibrated',
           'launches_per_step': 10, 'read_mb_per_launch': read_mb, 'write_mb_per_launch': write_mb,
           'hbm_bytes_per_launch': (read_mb + write_mb) * 1e6 if read_mb is not None and write_mb is not None else None,
           'algorithmic_mb_per_launch': 34.3, 'mfma_busy_frac': busy, 'l2_hit_rate': hit, 'by_kernel': kern}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
