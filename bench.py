#!/usr/bin/env python
"""Headline benchmark: captcha images/sec in TRAINING (forward + CTC + backward + all-reduce + clip + Adam) of the
VGG-7 + BiLSTM(256) + CTC CRNN at H=32, W=256, 10-char labels, batch 64 per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `roofline` is the dominant kernel (conv_k3 / conv_k2 / conv_halo: implicit-GEMM 3x3 convolution, MFMA-bound), timed
live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle (a torch-CPU restatement of the
reference TF1 graph — the TF reference itself cannot run here) on a bounded sample of the same workload.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, WIDTH, LABEL_LEN = 64, 256, 10
MFMA_BF16_PEAK = 2.5e15      # dense bf16, MI355X_MICROARCH.md
PEAK_CLOCK_MHZ = 2400            # 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz = 2.5 PFLOP/s
TRAIN_GFLOP_PER_IMG = 10.09


def synth_batches_varwidth(n_batches, seed, device):
    """BASELINE configs[3]: widths uniform in [80, 320], each batch right-padded with 0 to its max width rounded up to a
    multiple of 4 (gen.py:58-65), time_step_len = W_i // 4 - 1, labels of 4..10 characters."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n_batches):
        widths = rng.randint(80, 321, BATCH)
        wmax = int(np.ceil(widths.max() / 4.0) * 4)
        x = rng.rand(BATCH, wmax, 32).astype(np.float32)
        for i, w in enumerate(widths):
            x[i, w:] = 0.0
        ll = rng.randint(4, 11, BATCH).astype(np.int32)
        labels = rng.randint(1, 63, int(ll.sum())).astype(np.int32)
        sl = (widths // 4 - 1).astype(np.int32)
        out.append(tuple(torch.from_numpy(a).to(device) for a in (x, labels, ll, sl)))
    return out


def synth_batches(n_batches, seed, device):
    rng = np.random.RandomState(seed)
    out = []
    T = WIDTH // 4 - 1
    for _ in range(n_batches):
        x = torch.from_numpy(rng.rand(BATCH, WIDTH, 32).astype(np.float32)).to(device)
        labels = torch.from_numpy(rng.randint(1, 63, BATCH * LABEL_LEN).astype(np.int32)).to(device)
        ll = torch.full((BATCH,), LABEL_LEN, dtype=torch.int32, device=device)
        sl = torch.full((BATCH,), T, dtype=torch.int32, device=device)
        out.append((x, labels, ll, sl))
    return out


def pick_pmc_summary(profiles_dir, workload, build_id):
    """The whole-step counter summary (tools/pmc_step_summary.py) bench.py may quote for `workload` on the library build `build_id`:
    several rounds' summaries lie side by side in profiles/ — the one taken on this workload AND this build wins (the last by name if
    there are several); without one, the last by name is returned so that the refusal can name what it found.  (None, None): no file."""
    cands = sorted(glob.glob(os.path.join(profiles_dir, "r*_pmc_step_%s.json" % workload)))
    if not cands:
        return None, None
    loaded = [(c, json.load(open(c))) for c in cands]
    match = [cp for cp in loaded if cp[1].get("workload") == workload and cp[1].get("build_id") == build_id]
    return (match or loaded)[-1]


def conv_roofline(eng, device, workload):
    """Every launch of the dominant kernel (conv_k3 / conv_k2 / conv_halo, as the dispatcher picks them: the 3x3 SAME convolutions of the workload that was timed — forward
    and data gradient) timed with HIP events on the launch stream; achieved = sum(algorithmic flop) / sum(time).
    The layer shapes are read from the plan the timed steps ran (the widest one for variable-width batches)."""
    from lstm_ctc_ocr_amd import ops
    sp = max(eng.plans.values(), key=lambda p: p.W)
    shapes, first = [], True
    for op in eng.ops:
        if getattr(op, 'kind', None) == '3x3':
            (N, W, H, Ci), _ = sp.shape[op.key]
            pool = None
            if op.pool_after is not None and op.pool_after.key in getattr(sp, 'fused_pools', ()):
                pool = (op.pool_after.kw_t, op.pool_after.kh_f)          # the step's launch writes the max-pool too: time that form
            # batch-norm layers: the forward launch also leaves the statistics partials behind, and the data gradient INTO a batch-norm + ReLU
            # producer also takes that layer's backward sums (conv_k3b) — the probe times the launches the step runs
            fwd_stats = bool(getattr(sp, 'bn_stat_rows', {}).get(op.key, 0))
            dgrad_bnb = bool(getattr(sp, 'bn_bwd_rows', {}).get(op.prev.key, 0)) if hasattr(op, 'wdgrad') else False
            shapes.append((N, W, H, Ci, op.co, hasattr(op, 'wdgrad'), pool, fwd_stats, dgrad_bnb))
    tot_fl, exe_fl, tot_t, n_launch = 0.0, 0.0, 0.0, 0
    per_launch = []               # one record per launch: what, kernel, K steps, us, fraction of peak, stamped clock, matrix-pipe occupancy at that clock
    plain_t = 0.0                 # the same launches with the batch-norm work taken out of their write-outs again (secondary figure)
    # shader clock the kernel really runs at: workgroup 0 stamps {shader-clock counter, 100 MHz wall clock} at entry and exit
    from lstm_ctc_ocr_amd import _native as nat
    clk = torch.zeros(8, dtype=torch.int64, device=device)
    nat.call("ocr_conv_halo_clock_debug", clk.data_ptr())
    clk_cycles = clk_ticks = 0.0
    for (N, W, H, Ci, Co, has_dgrad, pool, fwd_stats, dgrad_bnb) in shapes:
        x = torch.randn(N, W, H, Ci, device=device).to(torch.bfloat16)
        y = torch.randn(N, W, H, Co, device=device).to(torch.bfloat16)
        wf = (torch.randn(Co, 3, 3, Ci, device=device) * 0.05).to(torch.bfloat16)
        wd = (torch.randn(Ci, 3, 3, Co, device=device) * 0.05).to(torch.bfloat16)
        b = torch.zeros(Co, device=device)
        oy, ox = torch.empty_like(y), torch.empty_like(x)
        if pool is not None:
            pooled = torch.empty(N, W // pool[0], H // pool[1], Co, device=device, dtype=torch.bfloat16)
            fns = [(lambda: ops.conv3x3_relu_pool(x, wf, oy, pooled, b, pool[0], pool[1]), ops.conv3x3_kernel_choice(N, W, H, Ci, Co, pool=pool))]
        elif fwd_stats:
            part = ops.bn_workspace(N * W * H, Co, device)
            fns = [(lambda: ops.conv3x3_stats(x, wf, oy, part, bias=b), ops.conv3x3_kernel_choice(N, W, H, Ci, Co, relu=False))]
        else:
            fns = [(lambda: ops.conv3x3(x, wf, out=oy, bias=b, relu=True), ops.conv3x3_kernel_choice(N, W, H, Ci, Co))]
        if has_dgrad and dgrad_bnb:
            part2 = ops.bn_workspace(N * W * H, Ci, device)
            mean, rstd = torch.zeros(Ci, device=device), torch.ones(Ci, device=device)
            fns.append((lambda: ops.conv3x3_dgrad_bnbwd(y, wd, ox, x, x, mean, rstd, part2),
                        ops.conv3x3_kernel_choice(N, W, H, Co, Ci, bias=False, relu=False, mask=True)))
        elif has_dgrad:
            fns.append((lambda: ops.conv3x3(y, wd, out=ox, mask=x), ops.conv3x3_kernel_choice(N, W, H, Co, Ci, bias=False, relu=False, mask=True)))
        plain = {}
        if fwd_stats:
            plain[0] = lambda: ops.conv3x3(x, wf, out=oy, bias=b, relu=False)
        if has_dgrad and dgrad_bnb:
            plain[1] = lambda: ops.conv3x3(y, wd, out=ox)        # (before round 4 this launch carried neither the mask nor the sums)
        for fi, (fn, kname) in enumerate(fns):
            if fi in plain:
                for _ in range(3):
                    plain[fi]()
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
                for _ in range(10):
                    plain[fi]()
                p1.record()
                torch.cuda.synchronize()
                plain_dt = p0.elapsed_time(p1) * 1e-3 / 10
            for _ in range(3):
                fn()
            clk.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            dt_l = e0.elapsed_time(e1) * 1e-3 / 10
            tot_t += dt_l
            plain_t += plain_dt if fi in plain else dt_l
            fl_l = 2.0 * N * W * H * 9 * Ci * Co
            ex_l = fl_l * ((1.0 - 2.0 / (3.0 * H)) if kname.startswith("conv_k3") else 1.0)
            tot_fl += fl_l
            exe_fl += ex_l
            n_launch += 1
            c = clk.cpu().numpy()                       # last of the ten back-to-back launches
            mhz_l = None
            if c[3] > c[1]:
                clk_cycles += float(c[2] - c[0]); clk_ticks += float(c[3] - c[1])
                mhz_l = float(c[2] - c[0]) / float(c[3] - c[1]) * 100.0
            cin_l = Ci if fi == 0 else Co               # contraction channels of this launch (the data gradient contracts over C_out)
            per_launch.append({"what": "%dx%dx%d %d->%d %s" % (N, W, H, Ci, Co, "forward" if fi == 0 else "data gradient"), "kernel": kname,
                               "k_steps": 9 * cin_l // 64, "us": dt_l * 1e6, "gflop": fl_l / 1e9, "frac": fl_l / dt_l / MFMA_BF16_PEAK,
                               "shader_clock_mhz": mhz_l,
                               "executed_frac_of_peak_at_that_clock": (ex_l / dt_l / (MFMA_BF16_PEAK * mhz_l / PEAK_CLOCK_MHZ)) if mhz_l else None,
                               "_exe": ex_l, "_cout": Co if fi == 0 else Ci, "_m": N * W * H})
    nat.call("ocr_conv_halo_clock_debug", None)
    # Two-point decomposition t = fixed + k_steps x per_step for every kernel instance that runs at two depths on the same output tile grid
    # (conv3_1 / conv3_2 forward on conv_k3/A at H = 8: 18 and 36 steps; conv4_1 / conv4_2 forward at H = 4: 36 and 72; ...): the fixed cost per
    # launch (dispatch, prologue, K-half exchange, write-out) and the K loop's matrix-pipe occupancy at the stamped clock — the three factors of
    # DESIGN section 9 (loop occupancy x non-fixed share x clock / 2400), per launch instead of argued.
    groups = {}
    for r in per_launch:
        groups.setdefault((r["kernel"], r["what"].split(" ")[0], r["_cout"], r["what"].endswith("forward")), []).append(r)
    for key, rs in groups.items():
        ks = sorted(set(r["k_steps"] for r in rs))
        if len(rs) == 2 and len(ks) == 2:
            a, b = sorted(rs, key=lambda r: r["k_steps"])
            per_step = (b["us"] - a["us"]) / (b["k_steps"] - a["k_steps"])
            fixed = a["us"] - a["k_steps"] * per_step
            for r in (a, b):
                mhz_r = r["shader_clock_mhz"]
                # executed MFMA clocks per K step and SIMD: executed flop of the launch / steps / (1024 SIMDs x 1024 flop per clock)
                mfma_clk = r["_exe"] / r["k_steps"] / (1024.0 * 1024.0)
                r["two_point"] = {"fixed_us": fixed, "per_k_step_us": per_step, "non_fixed_share": 1.0 - fixed / r["us"],
                                  "loop_mfma_occupancy_at_clock": (mfma_clk / (per_step * mhz_r)) if (mhz_r and per_step > 0) else None,
                                  "clock_over_2400": (mhz_r / PEAK_CLOCK_MHZ) if mhz_r else None}
    for r in per_launch:
        for k in ("_exe", "_cout", "_m"):
            r.pop(k)
    ach = tot_fl / tot_t
    mhz = clk_cycles / clk_ticks * 100.0 if clk_ticks else None
    # HBM bytes per launch and matrix-pipe occupancy of the same kernels from the separate rocprofv3 --pmc passes over THIS script
    # (tools/prof_step_pmc.sh -> profiles/rNN_pmc_step.json; FETCH_SIZE doubled: the guide's gfx950 correction).  The file names the
    # commit it was taken at and its kernel symbols: numbers whose symbols are not in the library loaded now are refused.
    traffic, mfma_busy, src, pmc_commit, pmc_error, pmc_build, pmc_clock, pmc_avg_us = None, None, None, None, None, None, None, None
    mfma_window, pmc_own_clock = None, None
    build_id = nat.build_id()
    try:
        build_commit = open(os.path.join(ROOT, ".build_commit")).read().strip()
    except OSError:
        build_commit = None
    # the newest profile of THIS workload; it must name the workload and the build id (source hash) of the library that is loaded now —
    # anything else is refused, not borrowed (the varwidth / deep lines used to print the headline's counters)
    path, pm = pick_pmc_summary(os.path.join(ROOT, "profiles"), workload, build_id)
    if pm is None:
        pmc_error = "no profiles/r*_pmc_step_%s.json for this workload" % workload
    else:
        src = os.path.basename(path)
        pmc_commit, pmc_build = pm.get("commit"), pm.get("build_id")
        conv = [k for k in pm.get("kernels", []) if k["symbol"].startswith(("_Z16conv_halo_kernel", "_Z14conv_k2_kernel", "_Z14conv_k3_kernel", "_Z15conv_k3w_kernel", "_Z15conv_k3b_kernel", "_Z14conv_ws_kernel"))]
        if pm.get("workload") != workload:
            pmc_error = "%s was taken on workload %r, this run is %r" % (src, pm.get("workload"), workload)
        elif pmc_build != build_id:
            pmc_error = "%s was taken on build %s (commit %s), the loaded library is build %s (commit %s)" % (src, pmc_build, pmc_commit, build_id, build_commit)
        elif not conv:
            pmc_error = "no convolution kernels in %s" % src
        else:
            nl = float(sum(k["launches"] for k in conv))
            tm = float(sum(k["launches"] * k["avg_us"] for k in conv))
            if all("read_mb" in k and "write_mb" in k for k in conv):
                traffic = sum(k["launches"] * (k["read_mb"] + k["write_mb"]) for k in conv) / nl * 1e6
            # matrix-pipe occupancy over the kernels' OWN cycles (busy cycles / (1024 SIMDs x duration x the shader clock stamped inside the
            # convolution launches of the counter pass)); the GRBM-window quotient of rounds 3-4 is kept under its own name (a lower bound)
            if all("mfma_busy_frac_own_cycles" in k for k in conv):
                mfma_busy = sum(k["launches"] * k["sq_pass_avg_us"] * k["mfma_busy_frac_own_cycles"] for k in conv) / sum(k["launches"] * k["sq_pass_avg_us"] for k in conv)
                pmc_own_clock = conv[0].get("own_clock_mhz")
            wkey = "mfma_busy_frac_grbm_window" if all("mfma_busy_frac_grbm_window" in k for k in conv) else "mfma_busy_frac"
            if all(wkey in k for k in conv):
                mfma_window = sum(k["launches"] * k["avg_us"] * k[wkey] for k in conv) / tm
            if all("clock_mhz" in k for k in conv):
                pmc_clock = sum(k["launches"] * k["avg_us"] * k["clock_mhz"] for k in conv) / tm
            pmc_avg_us = tm / nl
    return {"bound": "mfma", "kernel": "conv_ws_kernel / conv_k3_kernel / conv_k3b_kernel / conv_k2_kernel / conv_halo_kernel (implicit-GEMM 3x3 SAME conv, forward + data gradient, with their fused epilogues: %d launches/step)" % n_launch,
            "achieved": ach / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK,
            "avg_launch_us": tot_t / n_launch * 1e6, "algorithmic_gflop_per_launch": tot_fl / n_launch / 1e9,
            "frac_plain_write_outs": tot_fl / plain_t / MFMA_BF16_PEAK,
            "fused_note": "`frac` times the launches AS THE STEP RUNS THEM: since round 4 the batch-norm layers' forward statistics and the ReLU mask + "
                          "backward sums of a batch-norm producer ride in these kernels' write-outs (they replaced separate HBM-bound passes: the step is "
                          "faster, these launches are longer); frac_plain_write_outs times the same convolutions without that work",
            "executed_flop_frac": exe_fl / tot_fl, "achieved_executed": exe_fl / tot_t / 1e12,
            "executed_note": "`achieved` counts 2*M*K*N of every launch (SURVEY 8d) including the SAME-padding taps the plane-layout kernels "
                             "(conv_k3 / conv_k3w) never issue: 2/(3H) of a layer's MFMAs (H=4: a sixth); `achieved_executed` counts only issued MFMAs",
            "per_launch": per_launch,
            "traffic": traffic, "mfma_busy_frac": mfma_busy, "pmc_shader_clock_mhz": pmc_own_clock,
            "mfma_busy_frac_grbm_window_lower_bound": mfma_window, "pmc_grbm_window_clock_mhz": pmc_clock, "pmc_avg_launch_us": pmc_avg_us, "pmc_commit": pmc_commit, "pmc_build_id": pmc_build, "build_id": build_id,
            "build_commit": build_commit, "pmc_error": pmc_error,
            "shader_clock_mhz": mhz, "frac_of_peak_at_that_clock": (ach / (MFMA_BF16_PEAK * mhz / PEAK_CLOCK_MHZ)) if mhz else None,
            "executed_frac_of_peak_at_that_clock": (exe_fl / tot_t / (MFMA_BF16_PEAK * mhz / PEAK_CLOCK_MHZ)) if mhz else None,
            "clock_note": "shader clock measured inside EVERY timed convolution launch (s_memtime / s_memrealtime of workgroup 0 of conv_k3 / conv_k3w / "
                          "conv_halo, time-weighted); `peak` is the guide's dense bf16 figure at %d MHz; executed_frac_of_peak_at_that_clock is the "
                          "matrix-pipe occupancy of THIS run (issued MFMA cycles / available SIMD cycles); mfma_busy_frac is the same quantity from "
                          "the separate counter run, over the kernels' own cycles at THAT run's stamped clock (pmc_shader_clock_mhz)" % PEAK_CLOCK_MHZ,
            "traffic_note": "HBM bytes per launch (PMC FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of the same kernels inside this script's train "
                            "step, from profiles/%s.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x shader clock), the clock "
                            "stamped (s_memtime / s_memrealtime) inside the convolution launches of the counter pass itself (bench.py --stamp-clock); the busy "
                            "cycles are exact (16 per issued MFMA: profiles/r04_mfma_busy_calibration.md).  mfma_busy_frac_grbm_window_lower_bound divides "
                            "by GRBM_GUI_ACTIVE / 8 instead — the window of a profiled 20-60 us dispatch is longer than the kernel "
                            "(pmc_grbm_window_clock_mhz = window / duration reads above the chip's 2400 MHz), kept for continuity with rounds 3-4 only" % src}


def wgrad_roofline(eng, device, workload):
    """The OTHER third of the convolution MFMA work: the 3x3 weight gradients of the timed workload (tf.gradients of network.py:166 at
    train.py:81), launched as the step launches them — one slab kernel per layer (wgrad9p / wgrad9, deferred form) in backward order and the merged
    slab reductions wherever the engine's flush policy puts them (OCR_W9_FLUSH_MB) — timed as ONE sequence with HIP events on the launch stream.
    achieved = sum 2 M 9 C_in C_out / time of the whole sequence, reductions included (they compute nothing and are charged to the family)."""
    from lstm_ctc_ocr_amd import ops
    from lstm_ctc_ocr_amd import _native as nat
    sp = max(eng.plans.values(), key=lambda p: p.W)
    layers = []
    for op in reversed(eng.ops):                    # backward order
        if getattr(op, 'kind', None) == '3x3':
            (N, W, H, Ci), _ = sp.shape[op.key]
            need = ops.conv3x3_wgrad_workspace_bytes(N, W, H, Ci, op.co)
            if need:
                layers.append((N, W, H, Ci, op.co, need))
    if not layers:
        return None
    bufs = []
    for (N, W, H, Ci, Co, need) in layers:
        bufs.append((torch.randn(N, W, H, Ci, device=device).to(torch.bfloat16), (torch.randn(N, W, H, Co, device=device) * 0.1).to(torch.bfloat16),
                     torch.zeros(3, 3, Ci, Co, device=device), torch.zeros(Co, device=device), torch.empty(need, dtype=torch.uint8, device=device)))
    tables = {}

    def flush(pend):
        raw = b''.join(j for j, _ in pend)
        ent = tables.get(raw)
        if ent is None:
            tab = np.frombuffer(raw, dtype=eng.W9_JOB_DTYPE).copy()
            start = 0
            for i, (_, nblk) in enumerate(pend):
                tab['block_start'][i] = start
                start += nblk
            ent = tables[raw] = (torch.from_numpy(tab.view(np.uint8).copy()).to(device), len(pend), start)
        ops.wgrad9_reduce_jobs(*ent)

    counts = {"slab_kernels": 0, "reductions": 0}

    def sequence(count=False):
        pend, pbytes = [], 0
        for (x, dy, dw, db, ws) in bufs:
            job, nblk = ops.conv3x3_wgrad_deferred(x, dy, dw, db, ws)
            if count:
                counts["slab_kernels"] += 1
            if job is None:
                continue
            pend.append((job, nblk)); pbytes += ws.numel()
            if eng.w9_flush_bytes and pbytes >= eng.w9_flush_bytes:
                flush(pend); pend, pbytes = [], 0
                if count:
                    counts["reductions"] += 1
        if pend:
            flush(pend)
            if count:
                counts["reductions"] += 1

    sequence(count=True)
    for _ in range(2):
        sequence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sequence()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 10
    fl = sum(2.0 * N * W * H * 9 * Ci * Co for (N, W, H, Ci, Co, _) in layers)
    slab_mb = sum(need for (*_, need) in layers) / 1e6
    nl = counts["slab_kernels"] + counts["reductions"]
    traffic = pmc_us = src = None
    path, pm = pick_pmc_summary(os.path.join(ROOT, "profiles"), workload, nat.build_id())
    if pm is not None and pm.get("workload") == workload and pm.get("build_id") == nat.build_id():
        ks = [k for k in pm.get("kernels", []) if k["symbol"].startswith(("_Z14wgrad9p_kernel", "_Z13wgrad9_kernel", "_Z25wgrad9_reduce_jobs_kernel"))]
        steps = max(1, min(k["launches"] for k in ks)) if ks else 1
        if ks and all("read_mb" in k and "write_mb" in k for k in ks):
            per_step_launches = sum(k["launches"] for k in ks) / float(steps)
            traffic = sum(k["launches"] * (k["read_mb"] + k["write_mb"]) for k in ks) / float(steps) / per_step_launches * 1e6
            pmc_us = sum(k["launches"] * k["avg_us"] for k in ks) / float(steps)
            src = os.path.basename(path)
    return {"kernel": "wgrad9p_kernel / wgrad9_kernel + wgrad9_reduce_jobs_kernel (3x3 weight gradients: %d slab launches + %d merged slab reductions per step)"
                      % (counts["slab_kernels"], counts["reductions"]),
            "achieved": fl / t / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / t / MFMA_BF16_PEAK,
            "avg_launch_us": t / nl * 1e6, "sequence_us": t * 1e6, "launches": nl, "algorithmic_gflop": fl / 1e9,
            "slab_mb_written_and_read_back": slab_mb, "traffic": traffic, "pmc_sequence_us_in_the_step": pmc_us, "pmc_source": src,
            "note": "timed stand-alone, back to back, on random operands of the timed plan's shapes; `traffic` = HBM bytes per launch of the same kernels "
                    "inside the train step (counter summary of this build; null without one)"}


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (fp32 torch-CPU restatement of the TF1 graph — TensorFlow itself cannot run here) on bounded samples:
    `value` = training steps of the headline shape (W = 256, 10 characters; batch 8 to stay within seconds);
    `c1_*`  = BASELINE configs[0] (4-character captcha, W = 88, batch 8): training steps, and forward + greedy decode."""
    from oracle import decode as odec
    from oracle import graph as og
    # 16 threads: the oracle's ops are small (8 images); with one thread per core of a 256-core host the same step was
    # measured 250x SLOWER (344 s instead of ~1.4 s) from thread oversubscription — `cores` reports what was really used
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))

    def train_rate(W, L, budget):
        n = 8
        rng = np.random.RandomState(0)
        x = torch.from_numpy(rng.rand(n, W, 32).astype(np.float32))
        labels = rng.randint(1, 63, n * L).astype(np.int32)
        ll = np.full(n, L, np.int32)
        sl = [W // 4 - 1] * n
        params, state = og.init_params(), {}
        params, *_ = og.train_step(params, state, (x, labels, ll, sl), 1e-4, 1e-5)      # untimed warm-up step
        t0, steps = time.time(), 0
        while True:
            params, *_ = og.train_step(params, state, (x, labels, ll, sl), 1e-4, 1e-5)
            steps += 1
            if time.time() - t0 > budget or steps >= 120:
                break
        return n * steps / (time.time() - t0), steps, params, x, sl

    v, steps, _, _, _ = train_rate(WIDTH, LABEL_LEN, budget_s)
    c1, c1_steps, params, x, sl = train_rate(88, 4, 5.0)
    t0, reps = time.time(), 0
    with torch.no_grad():
        while time.time() - t0 < 3.0:
            odec.greedy_decode(og.forward(params, x, sl).numpy(), np.asarray(sl))
            reps += 1
    fwd = 8 * reps / (time.time() - t0)
    return {"value": v, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d training steps of %d images (W=256, L=10) with the fp32 CPU oracle" % (steps, 8),
            "c1_train_images_per_sec": c1, "c1_forward_greedy_images_per_sec": fwd,
            "c1_sample": "BASELINE configs[0]: %d training steps / %d forward+greedy-decode passes of 8 images (W=88, L=4)" % (c1_steps, reps)}


def self_launch(n):
    """Re-run this command line as n ranks under `python -m torch.distributed.run` on a free local port and pass the exit code on.
    Refuses when the box shows fewer GPUs than ranks (except with OCR_DIST_BACKEND=gloo, the test transport that shares one GPU)."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if os.environ.get("OCR_DIST_BACKEND", "nccl") == "nccl" and n > visible:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, visible))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the live per-launch timing of the dominant kernel (profiling runs)")
    ap.add_argument("--batch", type=int, default=None,
                    help="images per GPU and step (default: the workload's own — 64, deep 32).  Any other value is a SIDE measurement: the line "
                         "says so in `metric` and `config` and is never the headline number")
    ap.add_argument("--stamp-clock", action="store_true",
                    help="workgroup 0 of every convolution launch of the run adds its lifetime in shader clocks and wall ticks to a device "
                         "block; the line carries conv_clock = {mhz, launches} (the counter passes of tools/prof_step_pmc.sh use it)")
    ap.add_argument("--workload", choices=["fixed", "varwidth", "deep"], default="fixed",
                    help="fixed = BASELINE configs[1] (the headline metric); varwidth = configs[3] (W in [80,320] padded per "
                         "batch); deep = configs[4] (ResNet-34-style extractor + 2 x BiLSTM(512 per direction), 96 classes, bs=32/GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:               # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world is the answer
        args.gpus = world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without the launcher: re-exec under torch.distributed.run, one process per GPU (VERDICT r3 item 7 —
        # the driver's documented form passes the launcher itself; this makes the bare form produce the same line instead of nothing)
        return self_launch(args.gpus)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one process per GPU: torch.distributed.run --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))
    # OCR_DIST_BACKEND=gloo is for the test-suite only (tests/test_gpu_bench_two_ranks.py: two ranks of this script sharing the one GPU
    # of a test box, gradients staged through the host by lstm_ctc_ocr_amd.dist); the driver's runs use RCCL ("nccl")
    backend = os.environ.get("OCR_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def reduce_(t, op):                 # small host-visible reductions of the line's own bookkeeping
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=op)
        return h.to(t.device)

    from lstm_ctc_ocr_amd.config import cfg
    from lstm_ctc_ocr_amd.engine import Engine
    from lstm_ctc_ocr_amd.models import get_network
    cfg.TRAIN.SOLVER, cfg.TRAIN.LEARNING_RATE, cfg.TRAIN.WEIGHT_DECAY = 'Adam', 1e-4, 1e-5     # lstm/lstm.yml
    global BATCH
    net_name = 'LSTM_train'
    if args.workload == 'deep':
        # "2 x BiLSTM(512)" = 512 units per direction = TRAIN.NUM_HID 1024 (the reference's bi_lstm halves it: network.py:104-105)
        cfg.NCLASSES, cfg.TRAIN.NUM_LAYERS, cfg.TRAIN.NUM_HID, BATCH, net_name = 96, 2, 1024, 32, 'RESNET_train'
    default_batch = BATCH
    if args.batch is not None:
        BATCH = args.batch
    eng = Engine(get_network(net_name), device=device, seed=cfg.RNG_SEED, use_graphs=not args.no_graphs)
    eng.setup_optimizer()
    # fixed and deep: W = 256 for every batch (what their config strings say); varwidth: W in [80, 320] padded per batch
    batches = (synth_batches_varwidth if args.workload == 'varwidth' else synth_batches)(8, cfg.RNG_SEED + rank, device)

    def step(i):
        x, labels, ll, sl = batches[i % len(batches)]
        eng.train_step(x, labels, ll, sl, fetch_loss=False)

    stamp = None
    if args.stamp_clock:
        from lstm_ctc_ocr_amd import _native as nat
        stamp = torch.zeros(8, dtype=torch.int64, device=device)
        nat.call("ocr_conv_halo_clock_debug", stamp.data_ptr())
    # steps the guarded optimiser launch DROPPED (a persistent LSTM hand-off wait expired: the update kernels return early) and expired waits it
    # saw — device counters (scalars[73], [74]), read outside the timed region.  A dropped step is work skipped inside the timed region: the line
    # proves there was none, or carries no `value`
    g0 = eng.guard_counters()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    g1 = eng.guard_counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    g2 = eng.guard_counters()
    dp_host = None
    if eng.dp_host_s[5]:
        n = float(eng.dp_host_s[5])
        dp_host = {k: round(v / n * 1e6, 1) for k, v in zip(("graph1_fwd_ctc_bwd_late", "allreduce_late_enqueue", "graph2_bwd_early",
                                                               "allreduce_early_enqueue_and_join", "graph3_clip_optimiser_repack"), eng.dp_host_s[:5])}
        dp_host["note"] = "host-side enqueue time per step and phase (us), rank 0, timed steps + warm-up"
    if world > 1:
        dt = float(reduce_(torch.tensor([dt], dtype=torch.float64, device=device), dist.ReduceOp.MAX).item())
    loss = eng.last_loss()
    dp_check = None
    if world > 1:
        # data-parallel self-check on the hardware the line was measured on: every rank applied the same all-reduced gradient, so
        # the replicas' parameters must be bit-identical; their data streams are rank-seeded, so their local losses must differ
        chk = torch.stack([eng.params.double().sum(), eng.params.double().abs().sum()])
        lo, hi = reduce_(chk.clone(), dist.ReduceOp.MIN), reduce_(chk.clone(), dist.ReduceOp.MAX)
        ls = torch.tensor([loss], dtype=torch.float64, device=device)
        llo, lhi = reduce_(ls.clone(), dist.ReduceOp.MIN), reduce_(ls.clone(), dist.ReduceOp.MAX)
        dp_check = {"replicas_bit_identical": bool(torch.equal(lo, hi)), "local_loss_min": float(llo.item()), "local_loss_max": float(lhi.item())}
    # the same loop the way lib/lstm/train.py:130,139 runs it — the loss is fetched (one host sync) after EVERY step; reported
    # beside `value`, never as `value`
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        x, labels, ll, sl = batches[i % len(batches)]
        eng.train_step(x, labels, ll, sl, fetch_loss=True)
    torch.cuda.synchronize()
    dt_fetch = time.perf_counter() - t1
    # ... and the way lstm_ctc_ocr_amd/train.py runs it by default: every loss is read, one iteration behind (OCR_LOSS_LAG=1)
    t2 = time.perf_counter()
    pending = None
    for i in range(args.steps):
        x, labels, ll, sl = batches[i % len(batches)]
        eng.train_step(x, labels, ll, sl, fetch_loss=False)
        h = eng.report_async()
        if pending is not None:
            eng.report_wait(pending)
        pending = h
    eng.report_wait(pending)
    torch.cuda.synchronize()
    dt_lag = time.perf_counter() - t2

    g3 = eng.guard_counters()
    guard = torch.tensor([g1[0] - g0[0], g2[0] - g1[0], g3[0] - g2[0], g1[1] - g0[1], g2[1] - g1[1], g3[1] - g2[1]], dtype=torch.float64, device=device)
    if world > 1:                       # (every rank drops the same steps; the time-out counts are per rank: the line carries the maximum)
        guard = reduce_(guard, dist.ReduceOp.MAX)
    guard = [int(v) for v in guard.tolist()]
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = BATCH * world * args.steps / dt
        line = {
            "metric": "captcha images/sec training (32x256, bs=64/GPU)", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("VGG-7 + BiLSTM(256) + CTC train step, H=32 W=256, 10-char labels, C=64, bs=64/GPU "
                                    "(BASELINE.json configs[1]), Adam lr 1e-4 wd 1e-5 clip 10") if args.workload == "fixed" else
                                   ("VGG-7 + BiLSTM(256) + CTC train step, H=32, W in [80,320] padded per batch, masked CTC, "
                                    "bs=64/GPU (BASELINE.json configs[3])") if args.workload == "varwidth" else
                                   ("ResNet-34-style extractor + 2 x BiLSTM(512 units per direction, NUM_HID=1024) + CTC, 96 classes, H=32 W=256, "
                                    "bs=32/GPU (BASELINE.json configs[4]; bf16 MFMA operands where BASELINE says fp16: same MFMA rate, fp32 accumulation)"),
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world, "hipgraph": not args.no_graphs},
            "final_loss": loss,
            "dropped_steps": {"warmup": guard[0], "timed": guard[1], "side_loops": guard[2]},
            "lstm_timeouts": {"warmup": guard[3], "timed": guard[4], "side_loops": guard[5]},
            "guard": ("device-guarded optimiser step (ocr_optim_step_guarded2): on" if eng._guard_on() else "off (OCR_LSTM_TIMEOUT_GUARD=0): a time-out raises at the next report"),
            "fake_comm": ({"cus": int(os.environ["OCR_FAKE_COMM_CUS"]), "us_per_25mb": float(os.environ.get("OCR_FAKE_COMM_US", "250")),
                           "lds_kb": int(os.environ.get("OCR_FAKE_COMM_LDS_KB", "96")), "fake_world": int(os.environ["OCR_FAKE_WORLD"]),
                           "note": "one-GPU emulation: every all-reduce = a doubling kernel + `cus` resident workgroups holding CUs for the time a "
                                   "ring all-reduce of that range would run, on the communication stream (lstm_ctc_ocr_amd/dist.py)"}
                          if os.environ.get("OCR_FAKE_WORLD") and int(os.environ.get("OCR_FAKE_COMM_CUS", "0") or 0) > 0 else None),
            "dp_schedule": (None if not (eng.world > 1 or eng.force_allreduce) else
                            "one hipGraph per step, both bucket all-reduces captured on the communication stream (OCR_DP_GRAPH=1)" if (eng.dp_graph and eng.use_graphs and eng.overlap_allreduce)
                            else "three hipGraphs per step, the two bucket all-reduces issued between them" if eng.overlap_allreduce
                            else "two hipGraphs per step, one all-reduce between them (OCR_OVERLAP_ALLREDUCE=0)"),
            "dp_check": dp_check, "dp_host_enqueue_us": dp_host,
            "with_loss_fetch_every_step": {"value": BATCH * world * args.steps / dt_fetch, "ms_per_step": dt_fetch / args.steps * 1e3},
            "with_loss_read_one_step_behind": {"value": BATCH * world * args.steps / dt_lag, "ms_per_step": dt_lag / args.steps * 1e3},
            "model_tflops_per_gpu": value / world * TRAIN_GFLOP_PER_IMG * 1e9 / 1e12,
        }
        if guard[1] or guard[4]:
            # work was skipped inside the timed region: no number
            line["value"] = None
            line["ms_per_step"] = None
            line.pop("model_tflops_per_gpu", None)
            line["error"] = ("%d of the %d timed steps were dropped on the device after %d expired persistent-LSTM hand-off waits: the timed region "
                             "did not do the work of %d steps — no value is reported" % (guard[1], args.steps, guard[4], args.steps))
        if stamp is not None:
            nat.call("ocr_conv_halo_clock_debug", None)
            c = stamp.cpu().numpy()
            line["conv_clock"] = {"mhz": (float(c[4]) / float(c[5]) * 100.0) if c[5] else None, "launches": int(c[6]),
                                  "note": "shader clocks / 100 MHz wall ticks of workgroup 0, summed over every convolution launch of this process"}
        if args.workload != "fixed":
            line["metric"] = "captcha images/sec training (%s workload)" % args.workload
            line.pop("model_tflops_per_gpu", None)
        if BATCH != default_batch:          # (after the workload's own metric string: ADVICE r5)
            line["metric"] += " — SIDE MEASUREMENT at bs=%d/GPU (not the configuration BASELINE.json names)" % BATCH
            line["config"]["workload"] += " — run at bs=%d/GPU instead of %d" % (BATCH, default_batch)
        # the timed number above is complete at this point: a failure in the two side measurements must not cost the line (it is
        # reported inside the line instead of being swallowed)
        if not args.no_roofline:
            try:
                line["roofline"] = conv_roofline(eng, device, args.workload)
                try:
                    line["roofline"]["wgrad"] = wgrad_roofline(eng, device, args.workload)
                    wg = line["roofline"]["wgrad"]
                    if wg:
                        # forward + data gradient + weight gradient of every 3x3 layer: all convolution MFMA work of the step
                        r = line["roofline"]
                        tot_fl = sum(p["gflop"] for p in r["per_launch"]) * 1e9
                        tot_t = sum(p["us"] for p in r["per_launch"]) * 1e-6 + wg["sequence_us"] * 1e-6
                        r["all_conv_mfma_work"] = {"achieved": (tot_fl + wg["algorithmic_gflop"] * 1e9) / tot_t / 1e12,
                                                   "frac": (tot_fl + wg["algorithmic_gflop"] * 1e9) / tot_t / MFMA_BF16_PEAK,
                                                   "us_per_step": tot_t * 1e6}
                except Exception as e:          # noqa: BLE001
                    line["roofline"]["wgrad"] = {"error": "%s: %s" % (type(e).__name__, e)}
                if BATCH != default_batch:      # the committed counter summaries belong to the workload's own batch size
                    line["roofline"].update({"traffic": None, "mfma_busy_frac": None, "mfma_busy_frac_grbm_window_lower_bound": None,
                                             "pmc_error": "counter summaries are taken at the workload's own batch size (%d), this run is bs=%d" % (default_batch, BATCH)})
            except Exception as e:              # noqa: BLE001
                line["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:          # noqa: BLE001
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
