"""Developer aid: do Gibbs launches from two host threads (two panel handles / streams) overlap on the device?"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=256); ap.add_argument("--reads", type=int, default=20000)
ap.add_argument("--threads", type=int, default=2)
a = ap.parse_args()
panel = make_synthetic_panel(K=50000, nSNPs=64000, seed=4916)
devs = [DevicePanel(panel) for _ in range(a.threads)]
for d in devs:
    d.set_device_share(a.threads)
ns = max(1, a.chains // 7)
samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=a.reads) for i in range(ns)]
rng = np.random.default_rng(0)
S = [samples[c % ns] for c in range(a.chains)]
which = [np.sort(rng.choice(panel.K, 600, replace=False)).astype(np.int32) + 1 for _ in range(a.chains)]
H0 = [rng.integers(1, 3, size=s.nReads).astype(np.int32) for s in S]
fr = [0] * a.chains
sr = rng.integers(0, 2**63, size=a.chains).astype(np.uint64); ss = rng.integers(0, 2**63, size=a.chains).astype(np.uint64)

def run(dev, out, i):
    t0 = time.time()
    forwardBackwardGibbsNIPT_batch(dev, S, which, H0, None, fr, None, seed_reads=sr, seed_shard=ss,
                                   return_hapProbs=False, return_genProbs=False)
    out[i] = (t0, time.time())

for rep in range(2):
    out = [None] * a.threads
    t0 = time.time()
    run(devs[0], out, 0)
    solo = time.time() - t0
    ths = [threading.Thread(target=run, args=(devs[i], out, i)) for i in range(a.threads)]
    t0 = time.time()
    for t in ths: t.start()
    for t in ths: t.join()
    both = time.time() - t0
    print(f"rep {rep}: one thread {solo:.2f}s; {a.threads} threads concurrently {both:.2f}s "
          f"(intervals {[(round(s - t0, 2), round(e - t0, 2)) for s, e in out]})", flush=True)
