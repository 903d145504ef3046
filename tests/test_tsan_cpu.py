"""The product's host threading under ThreadSanitizer, and once more under AddressSanitizer + UBSan: csrc/impute.cpp,
csrc/bamrange.cpp, csrc/hostio.cpp and csrc/mspbwt.cpp compiled with g++ -fsanitize=... into tests/c/tsan_harness.cpp (trivial compute, the real threads: three host threads taking launch sets in
turn, helper threads, staggered start, fused tails, the sample source, loader threads settling files in order beside the call,
formatter pool, count sums; an unreadable file in the middle).  No report, exit code 0."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quilt_amd", "csrc")


def _sanitizer_works(tmp_path, san):
    src = tmp_path / "probe.cpp"
    src.write_text("#include <thread>\nint main(){ std::thread t([]{}); t.join(); return 0; }\n")
    exe = tmp_path / "probe"
    r = subprocess.run(["g++", "-fsanitize=" + san, str(src), "-o", str(exe), "-pthread"], capture_output=True)
    return r.returncode == 0 and subprocess.run([str(exe)], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0")).returncode == 0


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_threads_are_clean_under_the_sanitizers(tmp_path, small_panel, san):
    if shutil.which("g++") is None or not _sanitizer_works(tmp_path, san):
        pytest.skip(f"no working g++ -fsanitize={san} here")
    from quilt_amd.synth import make_synthetic_sample
    from tests import bamutil
    exe = tmp_path / "tsan_harness"
    stubs = tmp_path / "stubs.o"
    subprocess.run(["gcc", "-c", os.path.join(ROOT, "tests", "c", "tsan_stubs.c"), "-o", str(stubs)], check=True)
    build = subprocess.run(["g++", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-g", "-O1" if san == "thread" else "-O0", "-std=c++17", os.path.join(CSRC, "impute.cpp"),
                            os.path.join(CSRC, "bamrange.cpp"), os.path.join(CSRC, "hostio.cpp"), os.path.join(CSRC, "mspbwt.cpp"),
                            os.path.join(ROOT, "tests", "c", "tsan_harness.cpp"), str(stubs), "-lz", "-pthread", "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    # the BAM side: eleven files of one small panel, two of them without reads (one in the middle, one last)
    panel = small_panel
    rng = np.random.default_rng(5)
    T = panel.nSNPs
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(T)]
    ref, alt = [a for a, _ in alleles], [b for _, b in alleles]
    header = [("chr20", int(panel.L[-1]) + 1000)]
    paths = []
    for i in range(9):
        s = make_synthetic_sample(panel, seed=900 + i, n_reads=120 + 10 * i)
        p = str(tmp_path / f"t{i}.bam")
        bamutil.write_bam(p, header, bamutil.sample_to_alignments(s, panel.L, ref, alt, rng))
        paths.append(p)
    for at in (4, len(paths) + 1):
        p = str(tmp_path / f"empty{at}.bam")
        bamutil.write_bam(p, header, [])
        paths.insert(at, p)
    # one coordinate-sorted file with its .bai, whose index the harness damages (QA_HARNESS_INDEXED)
    s_idx = make_synthetic_sample(panel, seed=990, n_reads=400)
    alns = sorted(bamutil.sample_to_alignments(s_idx, panel.L, ref, alt, rng), key=lambda a: a["pos"])
    indexed = str(tmp_path / "indexed.bam")
    bamutil.write_bam(indexed, header, alns, index=True, block=2048)
    grid = np.ascontiguousarray(panel.grid if panel.grid is not None else np.arange(T) // 32, dtype=np.int32)
    with open(tmp_path / "sites.bin", "wb") as f:
        f.write(np.int32(T).tobytes())
        f.write(np.ascontiguousarray(panel.L, dtype=np.int32).tobytes())
        f.write("".join(ref).encode())
        f.write("".join(alt).encode())
        f.write(grid.tobytes())
    (tmp_path / "scratch").mkdir()
    env = dict(os.environ, QA_HARNESS_SCRATCH=str(tmp_path / "scratch"), QA_HARNESS_INDEXED=indexed, TSAN_OPTIONS="halt_on_error=0 exitcode=66 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=0:exitcode=67",
               UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), str(tmp_path / "sites.bin")] + paths, capture_output=True, text=True, env=env, timeout=600)
    for mark in ("ThreadSanitizer", "AddressSanitizer", "runtime error:"):
        assert mark not in run.stderr, run.stderr[-6000:]
    assert run.returncode == 0, (run.returncode, run.stderr[-3000:])
    assert "tsan harness: ok" in run.stdout
