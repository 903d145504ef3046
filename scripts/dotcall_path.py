"""What an UNMODIFIED QUILT() gets through the `.Call` shim (shim/quilt_amd_shim.c): the reference's own loop shape -- per
sample, per Gibbs sample, per seek iteration ONE `rcpp_forwardBackwardGibbsNIPT` (qa_gibbs_batch with n_chain = 1) and, per
read label, ONE `Rcpp_haploid_dosage_versus_refs` (QUILT/R/functions.R:2614, :2034) -- from W worker processes per GPU
(mclapply's children, QUILT/R/quilt.R:692; each creates its own device context and panel handle,
qa_panel_set_device_share(W)).  Reported as samples/s beside the batched driver's figure (bench.py: `dotcall_path`).

One chain per launch uses 1 of 1 024 SIMD slots (a few waves with the multi-wave geometry) and one full-panel pass one of
256 compute units: this is the boundary's slow path, measured so that the difference to qa_impute_samples (one call per
sample range) is a number and not a guess.

  python scripts/dotcall_path.py --workers 16 --samples 2        # prints one JSON object
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class PerCallBackend:
    """quilt_amd.driver's backend interface answered with the single-call entry points, one chain / one pass at a time."""
    select_on_device = False

    def __init__(self, dev):
        self.dev = dev
        self.n_gibbs_calls = 0
        self.n_fullpass_calls = 0

    def make_gl_bound(self, gl, minGLValue, to_fix):
        from quilt_amd.reference_single import Rcpp_make_gl_bound
        Rcpp_make_gl_bound(gl, minGLValue, to_fix)

    def gibbs_batch(self, samples, which, starts, seed_reads, first_reads, seed_shards, **kw):
        from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
        kw.pop("return_hapProbs", None)
        kw.pop("return_hap_words", None)
        init = bool(kw.pop("gibbs_initialize_iteratively", False))
        out = []
        for s, w, h, sr, fr, ss in zip(samples, which, starts, seed_reads, first_reads, seed_shards):
            self.n_gibbs_calls += 1
            out += forwardBackwardGibbsNIPT_batch(self.dev, [s], [w], [h], None, [max(fr, 0)], None, seed_reads=[sr], seed_shard=[ss],
                                                  return_hapProbs=False, return_genProbs=False,
                                                  gibbs_initialize_iteratively=init and fr >= 0, **kw)
        return out

    def fullpass_reads_batch(self, samples, chain_sample, labels, want_dosage, want_top, cols, K_top_matches, minGLValue, top_width,
                             n_label=2):
        from quilt_amd.driver import make_gl_from_u_bq
        from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
        P = self.dev.panel
        T = P.nSNPs
        n_chain = len(chain_sample)
        n_thin = int((np.asarray(cols) >= 0).sum())
        dosage = np.zeros((n_chain, n_label, T))
        top = np.full((n_chain, n_label, n_thin, top_width), -1, dtype=np.int32)
        cnt = np.zeros((n_chain, n_label, n_thin), dtype=np.int32)
        for c in range(n_chain):
            s = samples[chain_sample[c]]
            per_base = np.repeat(labels[c], np.diff(s.read_ptr))
            for l in range(1, n_label + 1):   # functions.R:2014-2115: one call per read label
                sel = (per_base == l) & (s.bq != 0)
                gl = make_gl_from_u_bq(s.u[sel], s.bq[sel], T, minGLValue, self.make_gl_bound)
                best = [None] * n_thin
                self.n_fullpass_calls += 1
                Rcpp_haploid_dosage_versus_refs(self.dev, gl, dosage=dosage[c, l - 1], gammaSmall_cols_to_get=cols,
                                                K_top_matches=K_top_matches, best_haps_stuff_list=best,
                                                return_betaHat_t=False, return_dosage=bool(want_dosage[c]), return_gamma_t=False,
                                                get_best_haps_from_thinned_sites=True, always_normalize=False)
                for j, b in enumerate(best):
                    order = np.argsort(-b["top_matches_values"], kind="stable")   # everything_per_hap_rejig_haps
                    k = b["top_matches"][order][:top_width]
                    top[c, l - 1, j, : len(k)] = k
                    cnt[c, l - 1, j] = len(b["top_matches"])
        return dosage, top, cnt

    def fullpass_batch(self, gls, want_dosage, cols, K_top_matches):
        from quilt_amd.driver import HipBackend
        return HipBackend(self.dev).fullpass_batch(gls, want_dosage, cols, K_top_matches)

    def read_confidence_batch(self, samples, haps, maxDifferenceBetweenReads):
        from quilt_amd.gibbs_nipt import calculate_eMatRead_t_vs_haplotypes_batch
        out = []
        for s, h in zip(samples, haps):   # calculate_eMatRead_t_vs_haplotypes once per Gibbs sample (functions.R:1147-1157)
            out += calculate_eMatRead_t_vs_haplotypes_batch(self.dev, [s], [list(h)], maxDifferenceBetweenReads)
        return out


def _worker(args):
    w, a = args
    os.environ.setdefault("QA_HOST_THREADS", "2")
    from quilt_amd import native
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=a["K"], nSNPs=a["nsnps"], seed=4916)
    samples = [make_synthetic_sample(panel, seed=9000 + w * a["samples"] + i, n_reads=a["reads"]) for i in range(a["samples"])]
    native.check(native.lib().qa_set_device(a["device"]))
    dev = DevicePanel(panel)       # the worker's own context and handle, created after the fork (quilt.R:692)
    dev.set_device_share(a["workers"])
    if a["fp64"]:
        dev.set_dosage_precision(64)
    be = PerCallBackend(dev)
    drv = Driver(panel, be, DriverParams(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=1))
    drv.run(samples[:1], sample_offset=w * 1000)   # warm-up: arenas, code objects
    be.n_gibbs_calls = be.n_fullpass_calls = 0
    a["barrier"].wait()
    t0 = time.time()
    for i, s in enumerate(samples):
        drv.run([s], sample_offset=w * 1000 + 1 + i)
    t1 = time.time()
    dev.close()
    return dict(worker=w, t0=t0, t1=t1, gibbs_calls=be.n_gibbs_calls, fullpass_calls=be.n_fullpass_calls,
                host_seconds={k: round(v, 2) for k, v in drv.timing.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=16, help="worker processes on the GPU (mclapply's nCores)")
    ap.add_argument("--samples", type=int, default=2, help="timed samples per worker (after one warm-up sample)")
    ap.add_argument("--K", type=int, default=50000)
    ap.add_argument("--nsnps", type=int, default=64000)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--mixed", action="store_true", help="dosage passes with fp32 state (default: fp64, the headline's precision)")
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        bar = mgr.Barrier(a.workers)
        cfg = dict(K=a.K, nsnps=a.nsnps, reads=a.reads, samples=a.samples, workers=a.workers, device=a.device, fp64=not a.mixed,
                   barrier=bar)
        with ctx.Pool(a.workers) as pool:
            res = pool.map(_worker, [(w, cfg) for w in range(a.workers)])
    span = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    n = a.workers * a.samples
    print(json.dumps({
        "what": "the `.Call` shim's path: per sample, per Gibbs sample, per seek iteration one qa_gibbs_batch(n_chain = 1) and one "
                "qa_Rcpp_haploid_dosage_versus_refs per read label, from W worker processes sharing the GPU",
        "value": n / span, "unit": "samples/sec", "workers": a.workers, "samples": n, "seconds": round(span, 2),
        "gibbs_calls_per_sample": res[0]["gibbs_calls"] / a.samples, "fullpass_calls_per_sample": res[0]["fullpass_calls"] / a.samples,
        "seconds_per_sample_per_worker": round(float(np.mean([(r["t1"] - r["t0"]) / a.samples for r in res])), 2),
        "host_seconds_worker0": res[0]["host_seconds"], "K": a.K, "nSNPs": a.nsnps, "reads": a.reads,
        "dtype": "f64" if not a.mixed else "mixed"}))


if __name__ == "__main__":
    main()
