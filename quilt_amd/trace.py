"""Host-side span trace (diagnostic): which host thread was in which phase when.

Off unless ``QUILT_AMD_TRACE`` names an output file; then every ``span(name)`` appends (thread, name, start, end) and
``dump()`` writes them as JSON.  ``device:*`` spans are the native calls (GIL released, the GPU working for this thread);
the rest is host work between them.  ``scripts/host_trace_states.py`` turns a dump into the fractions of time with no
thread / one thread / several threads inside a native call.
"""
from __future__ import annotations

import json
import os
import threading
import time
from contextlib import contextmanager

_PATH = os.environ.get("QUILT_AMD_TRACE", "")
_spans = []
_lock = threading.Lock()


def enabled() -> bool:
    return bool(_PATH)


@contextmanager
def span(name: str):
    if not _PATH:
        yield
        return
    t0 = time.perf_counter()
    try:
        yield
    finally:
        t1 = time.perf_counter()
        with _lock:
            _spans.append((threading.current_thread().name, name, t0, t1))


def add(name: str, t0: float, t1: float):
    """A span from two ``time.perf_counter()`` readings the caller already took."""
    if _PATH:
        with _lock:
            _spans.append((threading.current_thread().name, name, t0, t1))


def mark(name: str):
    """A zero-length span (an instant)."""
    if _PATH:
        t = time.perf_counter()
        with _lock:
            _spans.append((threading.current_thread().name, name, t, t))


def dump():
    if _PATH:
        with _lock:
            out = list(_spans)
        with open(_PATH, "w") as f:
            json.dump(out, f)
