"""Developer aid behind tests/test_sum_order_gpu.py: the seed sweeps of scripts/check_quick_start_seeds.py (ten driver seeds on the
quick-start-shaped panel, 27 % duplicated haplotypes) and scripts/check_seed_lists.py (24 seeds on the K = 5 000 panel) through the
native loop in BOTH modes of the full-panel passes -- production (block-wide tree sums) and validation (qa_panel_set_sum_order: the
reference's order) -- against the CPU pipeline, with the sampler's own noise floor (two CPU runs of the same sample under different
driver seeds) and r2 against the truth beside them.
    gpurun --timeout 900 -- 'python scripts/check_sum_order_seeds.py'
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.sum_order_cases import CASES, run_case

for name in CASES:
    t = time.time()
    rows = run_case(name, verbose=True)
    n = len(rows)
    div = [r for r in rows if not r["prod_labels_identical"]]
    print(f"== {name}: {n} seeds in {time.time() - t:.1f} s; validation mode: labels identical on {sum(r['val_labels_identical'] for r in rows)}/{n},"
          f" dosage bit-identical on {sum(r['val_dosage_identical'] for r in rows)}/{n}; production mode parts from the CPU path on {len(div)}/{n}")
    if div:
        print("   diverging seeds: r2(GPU, CPU) min %.5f median %.5f | noise floor r2(CPU seed a, CPU seed b) min %.5f median %.5f" % (
            min(r["prod_r2_vs_cpu"] for r in div), np.median([r["prod_r2_vs_cpu"] for r in div]),
            min(r["floor_r2"] for r in rows), np.median([r["floor_r2"] for r in rows])))
    d = np.array([r["prod_r2_truth"] - r["cpu_r2_truth"] for r in rows])
    f = np.array([r["cpu2_r2_truth"] - r["cpu_r2_truth"] for r in rows])
    print("   r2 vs truth, GPU - CPU: mean %+.5f sd %.5f min %+.5f | CPU other seed - CPU: mean %+.5f sd %.5f min %+.5f" % (
        d.mean(), d.std(), d.min(), f.mean(), f.std(), f.min()))
