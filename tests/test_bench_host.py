"""Host-side logic of bench.py that needs no GPU: which counter summary a bench line may quote, and that the committed summaries of the
final build are the ones it would pick (the library's source hash is what they are pinned on)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counter_summary_of_the_loaded_build_wins(tmp_path):
    b = _bench()
    mk = lambda name, wl, bid: (tmp_path / name).write_text(json.dumps(dict(workload=wl, build_id=bid, kernels=[])))
    assert b.pick_pmc_summary(str(tmp_path), "fixed", "aaaa") == (None, None)
    mk("r04_final_pmc_step_fixed.json", "fixed", "old0")
    mk("r04_final2_pmc_step_fixed.json", "fixed", "new1")           # sorts BEFORE r04_final_ by name ('2' < '_'): name order is not age
    mk("r04_final2_varwidth_pmc_step_varwidth.json", "varwidth", "new1")
    path, pm = b.pick_pmc_summary(str(tmp_path), "fixed", "new1")
    assert os.path.basename(path) == "r04_final2_pmc_step_fixed.json" and pm["build_id"] == "new1"
    path, pm = b.pick_pmc_summary(str(tmp_path), "fixed", "old0")
    assert os.path.basename(path) == "r04_final_pmc_step_fixed.json"
    # no summary of this build: the last by name is handed back so the refusal can name it (conv_roofline then reports pmc_error)
    path, pm = b.pick_pmc_summary(str(tmp_path), "fixed", "zzzz")
    assert os.path.basename(path) == "r04_final_pmc_step_fixed.json" and pm["build_id"] != "zzzz"
    # a summary of another workload is never picked, whatever its build
    path, pm = b.pick_pmc_summary(str(tmp_path), "varwidth", "new1")
    assert pm["workload"] == "varwidth"
    assert b.pick_pmc_summary(str(tmp_path), "deep", "new1") == (None, None)


def test_committed_counter_summaries_belong_to_the_sources_in_the_tree():
    """profiles/ holds, for each of the three bench workloads, a whole-step counter summary taken on a library built from EXACTLY the
    sources under lstm_ctc_ocr_amd/csrc now (sha256, `ocr_build_id`): a kernel edit without a new counter run fails here, on the CPU,
    instead of printing `pmc_error` in the round's bench line."""
    from lstm_ctc_ocr_amd import _native as nat
    b = _bench()
    bid = nat.source_build_id()
    for wl in ("fixed", "varwidth", "deep"):
        path, pm = b.pick_pmc_summary(os.path.join(ROOT, "profiles"), wl, bid)
        assert pm is not None and pm.get("build_id") == bid and pm.get("workload") == wl, (wl, path, pm and pm.get("build_id"), bid)


def test_counter_summary_divides_busy_cycles_by_the_kernels_own_cycles(tmp_path):
    """tools/pmc_step_summary.py (round 5): mfma_busy_frac_own_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the kernel's duration IN THE PASS THAT
    COUNTED IT x the shader clock bench.py --stamp-clock reported for that pass); the GRBM-window quotient keeps its own name; non-convolution
    kernels get no own-cycles figure (their launches are not stamped)."""
    import subprocess
    import sys
    txt = tmp_path / "pmc.txt"
    txt.write_text(
        "_Z14conv_ws_kernelILi16ELi16ELi1ELb0EEv6WsArgs.kd launches=11 avg_us=30.0\n"
        "    FETCH_SIZE avg 1000.0\n"
        "_Z14conv_ws_kernelILi16ELi16ELi1ELb0EEv6WsArgs.kd launches=11 avg_us=32.0\n"
        "    SQ_VALU_MFMA_BUSY_CYCLES avg 2.0e7\n"
        "    GRBM_GUI_ACTIVE avg 8.0e5\n"
        "_Z18adam_update_kernelPfPKfS_S_lffffPdllf.kd launches=11 avg_us=33.0\n"
        "    SQ_VALU_MFMA_BUSY_CYCLES avg 0.0\n"
        "    GRBM_GUI_ACTIVE avg 9.0e5\n")
    log = tmp_path / "pmc3.log"
    log.write_text('noise\n{"metric": "x", "conv_clock": {"mhz": 2000.0, "launches": 110}}\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_step_summary.py"), str(txt), ROOT, "fixed", str(log)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-800:]
    d = json.loads(out.stdout)
    assert d["workload"] == "fixed" and d["conv_clock"]["mhz"] == 2000.0
    k = {r["symbol"]: r for r in d["kernels"]}
    ws = k["_Z14conv_ws_kernelILi16ELi16ELi1ELb0EEv6WsArgs"]
    assert ws["sq_pass_avg_us"] == 32.0 and ws["avg_us"] == 30.0                      # the duration of the pass that held the counter
    assert abs(ws["mfma_busy_frac_own_cycles"] - 2.0e7 / (1024.0 * 32.0 * 2000.0)) < 1e-12
    assert abs(ws["mfma_busy_frac_grbm_window"] - 2.0e7 / (1024.0 * 8.0e5 / 8.0)) < 1e-12
    assert abs(ws["read_mb"] - 2.0 * 1000.0 * 1024 / 1e6) < 1e-9                      # FETCH_SIZE in KB, doubled (gfx950 correction)
    assert "mfma_busy_frac_own_cycles" not in k["_Z18adam_update_kernelPfPKfS_S_lffffPdllf"]
