"""tools/rocpd_pmc.py text (per-kernel counter averages of the four --pmc passes over bench.py) -> profiles/rNN_pmc_step.json: per kernel
HBM read bytes (FETCH_SIZE x 2: MI355X_MICROARCH.md — gfx950 reports half of wide coalesced reads; FETCH_SIZE / WRITE_SIZE are in KB),
write bytes (as reported, uncalibrated), matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) with kernel
cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), L2 hit rate, LDS conflict share, VALU instructions per MFMA — together with
the git commit and the kernel symbols, so that bench.py can refuse numbers taken from another build.
The json is pinned to the library that ran by `build_id` (ocr_build_id(): the hash of the sources the .so was built from) and names the
bench.py workload it was taken on: bench.py refuses numbers from another build or another workload.
Since round 5 the counter passes run bench.py with --stamp-clock: workgroup 0 of every convolution launch adds its own lifetime in shader
clocks and in 100 MHz wall ticks to a device block, bench.py prints the quotient (`conv_clock.mhz`), and the convolution rows carry
mfma_busy_frac_own_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration of the SAME pass x that clock) — the occupancy over the
kernel's own cycles.  The GRBM-window figure stays as mfma_busy_frac_grbm_window (a lower bound: the window of a profiled 20-60 us
dispatch is longer than the kernel, profiles/r04_mfma_busy_calibration.md).
usage: python tools/pmc_step_summary.py <txt> <repo root> [workload] [log of the SQ pass with bench.py's json line]"""
import json
import re
import subprocess
import sys


# What SQ_VALU_MFMA_BUSY_CYCLES adds up to per shader-clock cycle when every matrix pipe of the chip is saturated.  Rounds 1-3 assumed
# 1024 (one tick per SIMD and cycle); tools/mfma_busy_probe.py measures it (profiles/r04_mfma_busy_calibration.md).
MFMA_BUSY_UNITS_PER_CHIP_CYCLE = 1024.0


def main():
    kern, cur = {}, None
    for line in open(sys.argv[1]):
        m = re.match(r'(\S.*?)\s+launches=(\d+) avg_us=([\d.]+)', line)
        if m:
            cur = kern.setdefault(m.group(1), {'launches': int(m.group(2)), 'avg_us': float(m.group(3)), 'counters': {}, 'pass_us': {}})
            cur_us = float(m.group(3))
            continue
        m = re.match(r'\s+(\S+)\s+avg ([\d.e+-]+)', line)
        if m and cur is not None:
            cur['counters'][m.group(1)] = float(m.group(2))
            cur['pass_us'][m.group(1)] = cur_us          # the kernel's average duration in the pass that counted this
    try:
        commit = subprocess.check_output(['git', '-C', sys.argv[2], 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:
        commit = None
    try:      # the GPU box gets a snapshot without .git: the commit travels in .build_commit (git rev-parse --short HEAD > .build_commit before the call)
        commit = commit or open(sys.argv[2] + '/.build_commit').read().strip()
    except Exception:
        pass
    sys.path.insert(0, sys.argv[2])
    from lstm_ctc_ocr_amd import _native
    build_id = _native.build_id()
    workload = sys.argv[3] if len(sys.argv) > 3 else 'fixed'
    conv_clock = None           # bench.py --stamp-clock: {"mhz", "launches"} measured inside the convolution launches of the SQ pass
    if len(sys.argv) > 4:
        try:
            for line in open(sys.argv[4]):
                if line.startswith('{') and '"conv_clock"' in line:
                    conv_clock = json.loads(line).get('conv_clock')
        except OSError:
            pass
    CONV = ('_Z16conv_halo_kernel', '_Z14conv_k2_kernel', '_Z14conv_k3_kernel', '_Z15conv_k3w_kernel', '_Z15conv_k3b_kernel', '_Z14conv_ws_kernel')
    rows = []
    for name, k in kern.items():
        c = k['counters']
        g = lambda n: c.get(n)
        row = {'symbol': name.replace('.kd', ''), 'short': re.sub(r'^_Z\d+', '', name)[:60], 'launches': k['launches'], 'avg_us': k['avg_us']}
        if g('FETCH_SIZE') is not None: row['read_mb'] = 2.0 * g('FETCH_SIZE') * 1024 / 1e6
        if g('WRITE_SIZE') is not None: row['write_mb'] = g('WRITE_SIZE') * 1024 / 1e6
        if g('SQ_VALU_MFMA_BUSY_CYCLES') is not None and g('GRBM_GUI_ACTIVE'):
            row['mfma_busy_frac_grbm_window'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (MFMA_BUSY_UNITS_PER_CHIP_CYCLE * g('GRBM_GUI_ACTIVE') / 8.0)
        if g('SQ_VALU_MFMA_BUSY_CYCLES') is not None and conv_clock and conv_clock.get('mhz') and name.startswith(CONV):
            own = k['pass_us']['SQ_VALU_MFMA_BUSY_CYCLES'] * conv_clock['mhz']          # us x clocks per us
            row['mfma_busy_frac_own_cycles'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (MFMA_BUSY_UNITS_PER_CHIP_CYCLE * own)
            row['own_clock_mhz'] = conv_clock['mhz']; row['sq_pass_avg_us'] = k['pass_us']['SQ_VALU_MFMA_BUSY_CYCLES']
        if g('GRBM_GUI_ACTIVE'): row['clock_mhz'] = g('GRBM_GUI_ACTIVE') / 8.0 / k['avg_us']       # shader clocks per us of THIS (eager, one-at-a-time) run
        if g('SQ_INSTS_MFMA'): row['mfma_insts'] = g('SQ_INSTS_MFMA')
        if g('SQ_INSTS_MFMA'): row['valu_per_mfma'] = (g('SQ_INSTS_VALU') or 0.0) / g('SQ_INSTS_MFMA')
        if g('SQ_WAVE_CYCLES'):
            row['wait_any_frac'] = (g('SQ_WAIT_ANY') or 0.0) / g('SQ_WAVE_CYCLES'); row['wait_inst_frac'] = (g('SQ_WAIT_INST_ANY') or 0.0) / g('SQ_WAVE_CYCLES')
        if g('TCC_HIT_sum') is not None: row['l2_hit_rate'] = g('TCC_HIT_sum') / max(1.0, g('TCC_HIT_sum') + (g('TCC_MISS_sum') or 0.0))
        if g('SQ_LDS_IDX_ACTIVE'): row['lds_conflict_frac'] = (g('SQ_LDS_BANK_CONFLICT') or 0.0) / g('SQ_LDS_IDX_ACTIVE')
        rows.append(row)
    rows.sort(key=lambda r: -r['launches'] * r['avg_us'])
    print(json.dumps({'source': 'rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE | LDS + TCC; --kernel-trace only) over '
                                'bench.py --no-graphs --steps 3 --warmup 2 (5 eager train steps + the side loops of the line); FETCH_SIZE doubled '
                                '(gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE uncalibrated; mfma_busy_frac_own_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x the kernel\'s duration in that pass x the shader clock '
                                'stamped inside the convolution launches of that pass); mfma_busy_frac_grbm_window = SQ_VALU_MFMA_BUSY_CYCLES / (MFMA_BUSY_UNITS_PER_CHIP_CYCLE x GRBM_GUI_ACTIVE / 8): a lower bound, '
                                'the divisor calibrated with tools/mfma_busy_probe.py (profiles/r04_mfma_busy_calibration.md)',
                      'commit': commit, 'build_id': build_id, 'workload': workload, 'conv_clock': conv_clock, 'kernels': rows}, indent=1))


if __name__ == '__main__':
    main()
