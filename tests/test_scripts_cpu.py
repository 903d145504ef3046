"""The measurement helpers under scripts/ that bench.py's record depends on (no device needed)."""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


HEADER = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
          "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
          "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def _counter_csv(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(HEADER)
        for i, (kernel, grid, wg, value, t0, t1) in enumerate(rows):
            w.writerow([i, i, "Agent 2", 1, 1, 1, grid, 1, kernel, wg, 0, 0, 8, 0, 16, counter, value, t0, t1])


def test_pmc_summary_gives_bytes_per_launch_and_per_workgroup(tmp_path, monkeypatch):
    """Launches of different sizes in one profiled run: the per-workgroup figure is what bench.py scales by its own chains per
    launch (a Gibbs launch has one workgroup per chain)."""
    mod = _load("pmc_summary")
    gibbs = "void (anonymous namespace)::k_gibbs<10, 1>((anonymous namespace)::GibbsParams)"
    fetch, write, out = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "o.json"
    # one launch of 448 chains and one of 64 (64-thread workgroups): FETCH_SIZE in KiB units
    _counter_csv(fetch, "FETCH_SIZE", [(gibbs, 448 * 64, 64, 448 * 1000.0, 0, 10_000_000), (gibbs, 64 * 64, 64, 64 * 1000.0, 0, 5_000_000)])
    _counter_csv(write, "WRITE_SIZE", [(gibbs, 448 * 64, 64, 448 * 100.0, 0, 10_000_000), (gibbs, 64 * 64, 64, 64 * 100.0, 0, 5_000_000)])
    monkeypatch.setattr(sys, "argv", ["pmc_summary.py", str(fetch), str(write), str(out), "test command"])
    mod.main()
    k = json.load(open(out))["kernels"]["k_gibbs"]
    per_chain = 1000.0 * 1024 * 2 + 100.0 * 1024          # read side doubled (gfx950 wide reads), write side as counted
    assert k["launches"] == 2 and k["workgroups"] == 512
    assert abs(k["hbm_bytes_per_workgroup"] - per_chain) < 1e-6 * per_chain
    assert abs(k["hbm_bytes_per_launch"] - per_chain * 256) < 1e-6 * per_chain * 256


def test_trace_state_scripts_run_on_small_inputs(tmp_path, capsys, monkeypatch):
    # kernel trace: one Gibbs launch 0-10 ms, one full-panel kernel 8-12 ms, idle until a second Gibbs launch 20-30 ms
    kt = tmp_path / "kernel_trace.csv"
    with open(kt, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Workgroup_Size_X", "Grid_Size_X"])
        w.writerow(["KERNEL_DISPATCH", "k_gibbs<10, 1>", 0, 10_000_000, 64, 512 * 64])
        w.writerow(["KERNEL_DISPATCH", "k_fwd64<4, 3>", 8_000_000, 12_000_000, 512, 256 * 512])
        w.writerow(["KERNEL_DISPATCH", "k_gibbs<10, 1>", 20_000_000, 30_000_000, 64, 512 * 64])
    monkeypatch.setattr(sys, "argv", ["trace_states.py", str(kt)])
    _load("trace_states").main()
    out = capsys.readouterr().out
    assert "G-" in out and "GF" in out and "--" in out
    # host spans: two threads, one native call each, overlapping 1 s of 3
    ht = tmp_path / "host.json"
    json.dump([["T1", "device:gibbs", 0.0, 2.0], ["T2", "device:fullpass", 1.0, 3.0], ["T1", "finish", 2.0, 3.0]], open(ht, "w"))
    monkeypatch.setattr(sys, "argv", ["host_trace_states.py", str(ht)])
    _load("host_trace_states").main()
    out = capsys.readouterr().out
    assert "2:" in out and "1:" in out


def test_bench_names_its_workload_and_its_counter_summary():
    """bench.py: `config.workload` says which BASELINE configuration a run is (or that it is none), and every workload is priced on the
    committed counter summary of THAT workload (profiles/r05_pmc_traffic_<workload>.json), never on another's."""
    sys.path.insert(0, ROOT)
    import bench
    assert "configs[2]" in bench.workload_label("short", 50000, 128)
    assert "configs[1]" in bench.workload_label("short", 5000, 32)
    assert bench.workload_label("short", 64976, 128).startswith("not a BASELINE.json configuration") and "HRC" in bench.workload_label("short", 64976, 128)
    assert "configs[4]" in bench.workload_label("nipt", 50000, 128) and "configs[3]" in bench.workload_label("ont", 50000, 128)
    assert "QUILT2's default mode" in bench.workload_label("short", 50000, 128, True, True)
    assert bench.workload_label("short", 20000, 64).startswith("not a BASELINE.json configuration")
    want = {("short", 50000, 128, False, False): "r06_pmc_traffic.json", ("short", 5000, 32, False, False): "r05_pmc_traffic_configs1.json",
            ("nipt", 50000, 128, False, False): "r05_pmc_traffic_nipt.json", ("ont", 50000, 128, False, False): "r05_pmc_traffic_ont.json",
            ("short", 50000, 128, True, False): "r05_pmc_traffic_mspbwt.json", ("short", 50000, 128, True, True): "r05_pmc_traffic_quilt2_default.json"}
    for key, name in want.items():
        got = bench.pmc_file_for(*key)
        assert got is not None and os.path.basename(got) == name and os.path.exists(os.path.join(ROOT, got)), (key, got)
        d = json.load(open(os.path.join(ROOT, got)))
        assert "calibration" in d and "x 1024 x 2.0" in d["units"]   # the calibrated factors, from profiles/r05_fetch_calibration.json
    assert bench.pmc_file_for("nipt", 50000, 128, False, True) is None   # no summary of NIPT with rare + common: the line must say so
    cal = json.load(open(os.path.join(ROOT, "profiles", "r05_fetch_calibration.json")))["kernels"]
    for k in ("rd16", "rd8", "rd4"):
        assert abs(cal[k]["factor"] - 2.0) < 0.01
    assert abs(cal["rd8buf"]["factor"] - 2.0 * 4800 / 4864) < 0.01   # the sampler's column shape: x 2 on the 128-byte lines touched
    for k in ("wr16", "wr8", "wr8buf"):
        assert abs(cal[k]["factor"] - 1.0) < 0.01


def test_bench_whole_sample_cpu_baseline_small():
    """cpu_baseline_whole: whole samples, one per worker, through the entire pipeline on the CPU path; the first n_keep results are the
    r2 reference; a budget that runs out stops the workers and reports nothing rather than a partial sample."""
    sys.path.insert(0, ROOT)
    import bench
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=1000, nSNPs=640, seed=4916)
    params = dict(nGibbsSamples=2, n_seek_its=2, Ksubset=64, Knew=64, seed=1)
    work = [(make_synthetic_sample(panel, seed=1000 + i, n_reads=120), i) for i in range(3)]
    out, ref = bench.cpu_baseline_whole(panel, params, work, 2, 120, 1)
    assert out["whole_samples"] == 2 and out["cores"] == 2 and out["kind"] == "port" and out["value"] > 0
    assert ref["n"] == 1 and ref["ref"][0].dosage.shape == (panel.nSNPs,)
    out2, ref2 = bench.cpu_baseline_whole(panel, params, work, 2, 0.01, 1)
    assert out2 is None and ref2 is None
