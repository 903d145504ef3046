"""Several host threads per GPU, each with its own device panel handle, stream, arena and Driver.

The per-sample driver alternates device phases (Gibbs launches, full-panel passes) with host phases (marshalling, the
R-level logic between the native calls: haplotype re-selection, consensus labels).  One thread leaves the GPU idle
during its host phases; with two, one thread's kernels run while the other is on the host (the ctypes calls release the
GIL), and their Gibbs launches -- serial chains, one wave per SIMD -- fill the device together.  Samples are independent
and every chain owns its random stream, so the results do not depend on the split (reference: mclapply over
sampleRanges, quilt.R:691-692).
"""
from __future__ import annotations

import queue
import threading
from typing import Iterable, List, Optional, Tuple

from .driver import Driver, DriverParams, HipBackend, PhasingTail
from .native import DevicePanel, DeviceRareCommon
from .sharding import get_sample_range


class PairGate:
    """Optional meeting point of the host threads right before their Gibbs launches: a launch of 512 chains fills half the
    SIMDs and takes the chain's serial latency whatever runs beside it, so two launches that start together cost one launch
    time, two that start 0.3 s apart cost 1.3.  A thread waits at most ``timeout`` seconds for the others, then goes alone."""

    def __init__(self, n: int, timeout: float):
        self.n, self.timeout = n, timeout
        self.cv = threading.Condition()
        self.count, self.gen = 0, 0

    def wait(self) -> bool:
        with self.cv:
            gen = self.gen
            self.count += 1
            if self.count >= self.n:
                self.gen += 1
                self.count = 0
                self.cv.notify_all()
                return True
            ok = self.cv.wait_for(lambda: self.gen != gen, self.timeout)
            if not ok:
                self.count -= 1
            return ok

    def reset(self, n: int):
        with self.cv:
            self.n, self.count = n, 0
            self.gen += 1
            self.cv.notify_all()

    def leave(self):
        """A thread that has no launches left (uneven number of batches): the others stop waiting for it."""
        with self.cv:
            self.n = max(self.n - 1, 1)
            if self.count >= self.n:
                self.gen += 1
                self.count = 0
                self.cv.notify_all()


class DeviceWorkers:
    def __init__(self, panel, params: Optional[DriverParams] = None, n_workers: int = 2, rare_common=None,
                 cu_partition: bool = False, fp64_dosage: bool = False, split: str = "halves", gibbs_gate: float = 0.0,
                 pass_priority: bool = False, exclusive: bool = False, fuse_tails: bool = True,
                 split_remainder: bool = True):
        self.n = n_workers
        self.split_remainder = split_remainder   # split = "alternate": left-over batches are cut into one part per thread
        # the last batches' phasing rounds of all threads run together (driver.PhasingTail) instead of one after the other
        self.fuse_tails = fuse_tails and n_workers > 1
        # "halves": every batch is cut into one contiguous part per thread; "alternate": whole batches go to the threads in turn
        # (a thread's Gibbs launch then carries a whole batch's chains -- 1 024 at the defaults, one per SIMD -- instead of half)
        if split not in ("halves", "alternate"):
            raise ValueError("split must be 'halves' or 'alternate'")
        self.split = split
        self.devs = [DevicePanel(panel) for _ in range(n_workers)]
        for w, d in enumerate(self.devs):
            d.set_device_share(n_workers)
            if exclusive and n_workers > 1:   # launch sets take the device in turn, each with all of it
                d.set_exclusive(True)
            if fp64_dosage:   # dosage passes with fp64 state, as the reference (verification mode)
                d.set_dosage_precision(64)
            if pass_priority and n_workers > 1:   # full-panel calls ahead of the other threads' Gibbs launches
                d.set_pass_priority(True)
            if cu_partition and n_workers > 1:   # each thread's Gibbs chains on its own share of the CUs (measured slower: DESIGN.md 5)
                d.set_cu_partition(w, n_workers)
        self.drcs = [DeviceRareCommon(d, rare_common) if rare_common is not None else None for d in self.devs]
        self.drivers = [Driver(panel, HipBackend(d, r), params, rare_common=rare_common) for d, r in zip(self.devs, self.drcs)]
        if gibbs_gate > 0 and n_workers > 1:
            gate = PairGate(n_workers, gibbs_gate)
            for d in self.drivers:
                d.gibbs_gate = gate

    @property
    def timing(self):
        keys = self.drivers[0].timing.keys()
        return {k: sum(d.timing[k] for d in self.drivers) for k in keys}

    @property
    def n_gibbs_chain_calls(self) -> int:
        return sum(d.n_gibbs_chain_calls for d in self.drivers)

    def reset_timing(self):
        for d in self.drivers:
            d.timing = {k: 0.0 for k in d.timing}
            d.n_gibbs_chain_calls = 0

    def close(self):
        for r in self.drcs:
            if r is not None:
                r.close()
        for d in self.devs:
            d.close()

    def run_stream(self, batches: Iterable[Tuple[list, int]]):
        """Like Driver.run_stream: yields one list of SampleResult per batch, in order; every worker thread pipelines its own
        stream of batches.  ``split = "alternate"``: whole batches go to the threads in turn, the batches left over by the
        thread count are cut into one contiguous part per thread; ``"halves"``: every batch is cut that way."""
        batches = list(batches)
        n = self.n
        if self.split == "alternate":
            # a straggler batch alone on one thread at the end of a finite stream would expose that thread's host phases
            # between its launch sets, with the other threads idle: the left-overs are cut across the threads instead
            n_whole = len(batches) // n * n if self.split_remainder else len(batches)
            if n_whole == 0:
                n_whole = len(batches)
        else:
            n_whole = 0
        whole, rest = batches[:n_whole], batches[n_whole:]
        # `runs` of samples that are cut into one part per thread: (samples, offset, lengths of the caller's batches in it).
        # Left-over batches whose samples are consecutive (offset + length = the next one's offset) make ONE run -- two
        # left-over batches over four threads are four launches' worth, not eight small ones
        runs: List[Tuple[list, int, List[int]]] = []
        for samples, offset in rest:
            if self.split == "alternate" and runs and runs[-1][1] + len(runs[-1][0]) == offset:
                runs[-1] = (runs[-1][0] + list(samples), runs[-1][1], runs[-1][2] + [len(samples)])
            else:
                runs.append((list(samples), offset, [len(samples)]))

        def stream_of(w: int):
            out = whole[w::n]
            for samples, offset, _ in runs:
                lo, hi = get_sample_range(len(samples), n)[w]
                if hi > lo:
                    out.append((samples[lo:hi], offset + lo))
            return out

        outs = [queue.Queue() for _ in range(n)]
        if self.drivers[0].gibbs_gate is not None:
            self.drivers[0].gibbs_gate.reset(n)
        tail = PhasingTail(n) if self.fuse_tails else None
        # Staggered start: thread w prepares its first launch once thread w - 1 has handed its own to the device.  The
        # preparation (thousands of per-chain draws) is interpreter work: started together, the threads share the interpreter
        # lock and the first launch leaves when ALL of them are done, n times later than it needs to.
        started = [threading.Event() for _ in range(n)]
        for w, d in enumerate(self.drivers):
            d.phasing_tail = tail
            d._gate_left = False
            d.on_first_launch = started[w].set

        def work(w: int):
            try:
                if w > 0:
                    started[w - 1].wait(5.0)
                for res in self.drivers[w].run_stream(stream_of(w)):
                    outs[w].put(res)
            except BaseException as e:   # surfaced by the consumer
                if tail is not None:
                    tail.abort(e)
                outs[w].put(e)
            finally:
                started[w].set()
                if self.drivers[w].gibbs_gate is not None and not self.drivers[w]._gate_left:
                    self.drivers[w].gibbs_gate.leave()

        def take(w: int):
            res = outs[w].get()
            if isinstance(res, BaseException):
                raise res
            return res

        threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(n)]
        for t in threads:
            t.start()
        for i in range(len(whole)):
            yield take(i % n)
        for samples, _, lengths in runs:
            merged: List = []
            for w in range(n):
                lo, hi = get_sample_range(len(samples), n)[w]
                if hi > lo:
                    merged.extend(take(w))
            at = 0
            for n_b in lengths:     # back to the caller's batches
                yield merged[at:at + n_b]
                at += n_b
        for t in threads:
            t.join()
        for d in self.drivers:
            d.phasing_tail = None


class StubWorkers:
    """Test hook of ``bench.py --stub``: the DeviceWorkers surface without any device (or oracle) work -- every sample comes
    back with zero dosages.  Lets the multi-rank launch / rendezvous / timing / reporting path of the bench be exercised
    on a machine without GPUs; it measures nothing."""

    def __init__(self, panel, params: Optional[DriverParams] = None):
        self.panel = panel
        self.devs = []
        self.timing = {"gibbs": 0.0, "fullpass": 0.0, "host": 0.0, "consensus": 0.0, "finish": 0.0, "accumulate": 0.0,
                       "new_batch": 0.0}

    def reset_timing(self):
        self.timing = {k: 0.0 for k in self.timing}

    def close(self):
        pass

    def run_stream(self, batches):
        import numpy as np
        from .driver import SampleResult
        T = self.panel.nSNPs
        for samples, _ in batches:
            yield [SampleResult(np.zeros(T), np.zeros((3, T)), np.zeros((T, 2)), np.zeros(s.nReads, dtype=np.int32), 0)
                   for s in samples]


class NativeWorkers:
    """The DeviceWorkers surface over the native driver loop (``qa_impute_samples``, csrc/impute.cpp): the host threads, the
    pipelining of launch sets and the handling of the stream's end are C++ there; a stream of batches becomes ONE native call
    (the batches' samples must be consecutive: they are one sample range).  ``n_workers`` panel handles = host threads."""

    def __init__(self, panel, params: Optional[DriverParams] = None, n_workers: int = 3, fp64_dosage: bool = False,
                 exclusive: bool = True, fuse_tails: bool = True, rare_common=None):
        self.n = n_workers
        self.panel = panel
        self.params = (params or DriverParams()).resolved(panel.K)
        if self.params.impute_rare_common and rare_common is None:
            raise ValueError("impute_rare_common needs the panel's rare/common tables")
        self.fuse_tails = fuse_tails
        self.devs = [DevicePanel(panel) for _ in range(n_workers)]
        for d in self.devs:
            d.set_device_share(n_workers)
            if exclusive and n_workers > 1:
                d.set_exclusive(True)
            if fp64_dosage:
                d.set_dosage_precision(64)
        self.drcs = [DeviceRareCommon(d, rare_common) for d in self.devs] if rare_common is not None else []
        # (bench.py's stand-alone kernel timings drive single rounds through the Python statement of the loop)
        self.drivers = [Driver(panel, HipBackend(self.devs[0], self.drcs[0] if self.drcs else None), params, rare_common=rare_common)]
        self.stats = {}
        self.reset_timing()

    def reset_timing(self):
        self.timing = {"gibbs": 0.0, "fullpass": 0.0, "host": 0.0, "consensus": 0.0, "finish": 0.0, "accumulate": 0.0,
                       "new_batch": 0.0}
        self.n_gibbs_chain_calls = 0
        self.n_gibbs_launches = 0

    def close(self):
        from .native import lib
        lib().qa_impute_release_buffers()
        for r in self.drcs:
            r.close()
        for d in self.devs:
            d.close()

    def prepare(self, batches: Iterable[Tuple[list, int]]):
        """The stream's samples in the form the C ABI takes them (quilt_amd.impute.PreparedRange: reads back to back, parameter
        structs): what a caller that holds flat host buffers hands over.  ``run_stream(prepared)`` then starts from there."""
        from .impute import prepare_range
        batches = list(batches)
        if not batches:
            return None
        flat, at = [], batches[0][1]
        for samples, offset in batches:
            if offset != at:
                raise ValueError("NativeWorkers.run_stream takes consecutive batches (one sample range)")
            flat.extend(samples)
            at += len(samples)
        r = prepare_range(self.devs, flat, self.params, sample_offset=batches[0][1], samples_per_launch_set=len(batches[0][0]),
                          fuse_tails=self.fuse_tails, drcs=self.drcs)
        r.batch_sizes = [len(smp) for smp, _ in batches]
        return r

    def run_stream(self, batches):
        """``batches``: an iterable of (samples, offset) with consecutive offsets, or what ``prepare`` made of one."""
        from .impute import PreparedRange, run_prepared
        prep = batches if isinstance(batches, PreparedRange) else self.prepare(batches)
        if prep is None:
            return
        res, st = run_prepared(prep, return_stats=True)
        batches = [(res[:n], None) for n in prep.batch_sizes]   # (only the sizes are used below)
        self.stats = st
        for k in ("gibbs", "fullpass", "host", "consensus", "finish", "accumulate"):
            self.timing[k] += st["ms_" + k] / 1e3
        self.n_gibbs_chain_calls += st["gibbs_chain_calls"]
        self.n_gibbs_launches += st["gibbs_launches"]
        at = 0
        for samples, _ in batches:
            yield res[at:at + len(samples)]
            at += len(samples)
