"""Calibration of SQ_VALU_MFMA_BUSY_CYCLES.  ocr_mfma_busy_probe saturates the matrix pipe with an exactly known number
of v_mfma_f32_16x16x32_bf16 (16 clocks of a SIMD's pipe each at 1024 flop / clock / SIMD); this script launches it in several occupancies
and writes what was launched; run under rocprofv3 --pmc and joined with tools/rocpd_pmc.py --each it gives the counter's unit:

    python tools/mfma_busy_probe.py launch  gpurun_out/mfma_probe_launches.json                (inside rocprofv3 --pmc ... --kernel-trace)
    python tools/mfma_busy_probe.py report  gpurun_out/mfma_probe_launches.json  <rocpd_pmc --each output>

report prints, per configuration: counter / (MFMA instructions x 16), the busy fraction pmc_step_summary.py's formula gives, and the true
pipe occupancy (instructions x 16 clocks / (SIMDs x kernel duration x measured clock))."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = [(256, 64, 20000), (256, 256, 20000), (256, 512, 10000), (1024, 256, 5000), (2048, 512, 1250), (128, 256, 20000)]
REPS = 3


def launch(path):
    import torch
    from lstm_ctc_ocr_amd import _native as nat
    dev = torch.device('cuda', 0)
    out = torch.zeros(64, device=dev)
    clk = torch.zeros(8, dtype=torch.int64, device=dev)
    rows = []
    nat.call('ocr_mfma_busy_probe', out.data_ptr(), 256, 256, 2000, None, nat.stream())       # warm-up (code load, clocks)
    torch.cuda.synchronize()
    for (nb, th, it) in CONFIGS:
        for _ in range(REPS):
            clk.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nat.call('ocr_mfma_busy_probe', out.data_ptr(), nb, th, it, clk.data_ptr(), nat.stream())
            e1.record()
            torch.cuda.synchronize()
            c = clk.cpu().numpy()
            mhz = float(c[2] - c[0]) / max(1.0, float(c[3] - c[1])) * 100.0
            rows.append(dict(blocks=nb, threads=th, iters=it, n_mfma=nb * (th // 64) * it * 8, event_us=e0.elapsed_time(e1) * 1e3,
                             loop_clocks_wg0=int(c[2] - c[0]), shader_mhz=mhz))
    json.dump(rows, open(path, 'w'), indent=1)
    for r in rows:
        print(r)


def report(path, each):
    rows = json.load(open(path))
    disp = [json.loads(l) for l in open(each) if l.startswith('{') and 'mfma_busy_probe' in l][1:]       # [0] = warm-up launch
    assert len(disp) == len(rows), (len(disp), len(rows))
    print('| blocks x threads x iters | MFMA instr | kernel us | MHz | BUSY / (instr x 16) | BUSY / (instr x 4) | formula busy frac (old: / 1024 SIMDs) | true pipe occupancy | SQ_INSTS_MFMA / instr |')
    print('|---|---|---|---|---|---|---|---|---|')
    for r, d in zip(rows, disp):
        busy, gui = d.get('SQ_VALU_MFMA_BUSY_CYCLES'), d.get('GRBM_GUI_ACTIVE')
        n = r['n_mfma']
        cyc = gui / 8.0 if gui else None
        true_occ = n * 16.0 / (1024.0 * d['us'] * r['shader_mhz'])
        print('| %d x %d x %d | %d | %.1f | %.0f | %.4f | %.4f | %s | %.4f | %s |' % (
            r['blocks'], r['threads'], r['iters'], n, d['us'], r['shader_mhz'], busy / (n * 16.0), busy / (n * 4.0),
            ('%.4f' % (busy / (1024.0 * cyc))) if cyc else None, true_occ,
            ('%.4f' % (d['SQ_INSTS_MFMA'] / n)) if d.get('SQ_INSTS_MFMA') else None))


if __name__ == '__main__':
    if sys.argv[1] == 'launch':
        launch(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3])
