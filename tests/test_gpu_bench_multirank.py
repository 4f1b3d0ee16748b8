"""Dry run of bench.py's N > 1 code path on ONE GPU: two ranks (torch.distributed.run, as the driver launches it) share cuda:0
and exchange through gloo instead of RCCL.  Exercises what single-rank runs never touch: ExchangeAdam's chunked asynchronous
all-reduce + fused per-chunk Adam on the device, the deferred counts all-reduce, synchronised grid updates, the max-over-ranks
timing, the fixed-global-batch mode of configs[3]."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("extra", [[], ["--rays-per-iter", "4096"]])
def test_two_ranks_on_one_device(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--pretrain", "40", "--pool", "65536", "--dist-backend", "gloo", "--all-ranks-on-device0"] + extra
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2
    assert math.isfinite(out["value"]) and out["value"] > 0 and out["samples_per_sec"] > 0
    assert out["scaling"] == ("strong" if extra else "weak")
    if extra:
        assert out["config"]["rays_per_iter_per_gpu"] == 2048       # 4096 global rays / 2 ranks
    assert "other_loop" in out and out["other_loop"]["ms_per_step"] > 0
