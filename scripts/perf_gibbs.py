"""Developer aid: time the Gibbs kernels at production scale on the GPU."""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quilt_amd import native
from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=50000); ap.add_argument("--T", type=int, default=64000)
ap.add_argument("--chains", type=int, default=224); ap.add_argument("--reads", type=int, default=20000)
ap.add_argument("--Ks", type=int, default=600); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--init-iter", action="store_true"); ap.add_argument("--samples", type=int, default=0)
ap.add_argument("--nipt", action="store_true", help="three read labels, block Gibbs (ff = 0.2)")
a = ap.parse_args()
panel = make_synthetic_panel(K=a.K, nSNPs=a.T, seed=4916)
dev = DevicePanel(panel)
ns = a.samples or max(1, a.chains // 7)
samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=a.reads, **({"ff": 0.2} if a.nipt else {})) for i in range(ns)]
rng = np.random.default_rng(0)
S = [samples[c % ns] for c in range(a.chains)]
which = [np.sort(rng.choice(panel.K, a.Ks, replace=False)).astype(np.int32) + 1 for _ in range(a.chains)]
H0 = [rng.integers(1, 4 if a.nipt else 3, size=s.nReads).astype(np.int32) for s in S]
fr = [int(rng.integers(0, s.nReads)) for s in S]
sr = rng.integers(0, 2**63, size=a.chains).astype(np.uint64); ss = rng.integers(0, 2**63, size=a.chains).astype(np.uint64)
for r in range(a.reps):
    native.lib().qa_profile_reset()
    t0 = time.time()
    out = forwardBackwardGibbsNIPT_batch(dev, S, which, H0, None, fr, None, seed_reads=sr, seed_shard=ss,
                                         gibbs_initialize_iteratively=a.init_iter, return_genProbs=False,
                                         **({"ff": [0.2] * a.chains} if a.nipt else {}))
    wall = time.time() - t0
    res = []
    L = native.lib()
    L.qa_profile_name.restype = C.c_char_p
    slots = {}
    for k in range(L.qa_profile_count()):   # (the sampler's 256-register build has a slot of its own: "k_gibbs<10, 1, true>")
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        L.qa_profile_get(k, C.byref(ms), C.byref(n), C.byref(b))
        nm = L.qa_profile_name(k).decode()
        key = "gibbs" if nm.startswith("k_gibbs") else nm[2:]
        t = slots.setdefault(key, [0.0, 0.0])
        t[0] += ms.value; t[1] += b.value
    for name in ("ematread", "gibbs", "happrobs"):
        ms_v, b_v = slots.get(name, [0.0, 0.0])
        res.append(f"{name} {ms_v:.1f} ms ({b_v / 1e9 / max(ms_v, 1e-9) * 1e3:.0f} GB/s)")
    steps = 21 * (a.reads + panel.nGrids)
    print(f"rep {r}: wall {wall:.2f}s  " + "  ".join(res) + f"  -> {float(res[1].split()[1]) * 1e3 / steps:.3f} us/step/chain", flush=True)
print("labels changed in chain 0:", int((out[0]['H'] != H0[0]).sum()), "of", len(H0[0]))
