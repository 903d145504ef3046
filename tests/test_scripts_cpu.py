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
