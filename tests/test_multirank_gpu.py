"""The 8-rank run, rehearsed on the one GPU there is: `bench.py --gpus 8 --one-device` -- eight processes (torch.distributed.run,
gloo rendezvous) that each upload the panel, take an arena (QA_ARENA_FRACTION = 0.1), pin their transfer buffers, run three host
threads through the device gate of THEIR process and impute their own contiguous sample range on device 0 at the same time.  Every
step's results must equal, bit for bit, those of a 1-rank run over the same global steps (the random streams are keyed by the
global sample index).  It gives no scaling number; it finds what would otherwise fail in the first minute of the real 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--K", "5000", "--nsnps", "3200", "--reads", "800", "--batch", "4", "--no-cpu-baseline", "--r2-vs-cpu", "0", "--no-alone",
          "--dotcall", "0", "--precision", "fp64", "--fuse", "1"]


def _run(args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_eight_ranks_on_one_device_equal_one_rank(tmp_path):
    d8, d1 = str(tmp_path / "d8"), str(tmp_path / "d1")
    line8 = _run(["--gpus", "8", "--one-device", "--steps", "2", "--warmup", "0", "--digest", d8] + COMMON, {"QA_ARENA_FRACTION": "0.1"})
    assert line8["n_gpus"] == 8 and line8["value"] > 0 and "one_device_rehearsal" in line8
    assert line8["ranks"]["n"] == 8 and len(line8["ranks"]["elapsed_s"]) == 8
    # rank r's steps are the global steps 2 r, 2 r + 1: the same 16 steps from one rank
    _run(["--gpus", "1", "--steps", "16", "--warmup", "0", "--digest", d1] + COMMON, {})
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d8)) == sorted(f"step_{i}.sha256" for i in range(16))
    for n in names:
        assert open(os.path.join(d1, n)).read() == open(os.path.join(d8, n)).read(), n
