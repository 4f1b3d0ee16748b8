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


def _clean_env():
    """the launch styles under test must not inherit a rendezvous from whatever runs pytest"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("launch", ["torchrun", "self-spawn"])
@pytest.mark.parametrize("extra", [[], ["--rays-per-iter", "4096"]])
def test_two_ranks_on_one_device(extra, launch):
    """launch = torchrun: the driver's N > 1 command; self-spawn: `python bench.py --gpus 2` with no launcher around it
    (the shape of the driver's N = 1 command) starts its own ranks and still prints ONE line."""
    bench_args = ["--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "2", "--pretrain", "40", "--pool", "65536",
                  "--dist-backend", "gloo", "--all-ranks-on-device0"] + extra
    if launch == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + bench_args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=_clean_env())
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]            # stdout carries the JSON line and NOTHING else (no RCCL banner)
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2
    assert math.isfinite(out["value"]) and out["value"] > 0 and out["samples_per_sec"] > 0
    assert out["scaling"] == ("strong" if extra else "weak")
    if extra:
        assert out["config"]["rays_per_iter_per_gpu"] == 2048       # 4096 global rays / 2 ranks
    assert "other_loop" in out and out["other_loop"]["ms_per_step"] > 0
    assert len(out["ms_per_step_windows"]) == 2 and out["ms_per_step"] in out["ms_per_step_windows"]
    assert out["exchange_bytes"] == 4 * 4 * 128**3 and out["comm_ms_per_step"] >= 0 and out["comm_window_ms_per_step"] >= 0
    assert set(out["aux"]["exchange_modes"]) == {"allreduce", "rs_ag"}          # the other exchange mode was stepped too
    assert out["replicas"]["identical"] and len(out["replicas"]["fingerprints"]) == 2, out["replicas"]


def _device_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X: the first multi-GPU box runs it (VERDICT r5 item 8)")
@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_two_real_devices_rccl(mode):
    """the driver's N = 2 command over RCCL / xGMI, one rank per GPU, both exchange modes: one stdout line, the comm figures on it, and
    the replicas bit-identical after the timed steps (every rank's field parameters and occupancy grid: `replicas` on the line)"""
    bench_args = ["--gpus", "2", "--steps", "10", "--warmup", "3", "--windows", "1", "--pretrain", "100", "--pool", "262144",
                  "--exchange-mode", mode, "--no-cpu-baseline", "--no-profile", "--no-scene-sweep", "--no-rank-step"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + bench_args
    env = _clean_env()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # (the host driver only supports dmabuf IPC)
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 10 and math.isfinite(out["value"]) and out["value"] > 0
    assert out["comm_ms_per_step"] >= 0 and out["comm_window_ms_per_step"] > 0 and out["exchange_bytes"] == 4 * 4 * 128**3
    assert out["replicas"]["identical"] and len(out["replicas"]["fingerprints"]) == 2, out["replicas"]
    assert set(out["aux"]["exchange_modes"]) == {"allreduce", "rs_ag"}


def test_rccl_world_of_one():
    """the first RCCL initialisation of this repository: backend "nccl" with ONE rank, the step through ExchangeAdam's chunked
    all-reduce (launched from the gradient hook inside backward) + fused per-chunk Adam, comm figures on the line"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "4",
           "--warmup", "2", "--windows", "1", "--pretrain", "40", "--pool", "65536", "--no-cpu-baseline", "--no-aux", "--no-profile"]
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=_clean_env())
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]            # stdout carries the JSON line and NOTHING else (no RCCL banner)
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and math.isfinite(out["value"]) and out["value"] > 0
    assert out["exchange_bytes"] == 4 * 4 * 128**3 and out["comm_window_ms_per_step"] > 0


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_exchange_adam_on_rccl_matches_fused_adam(mode):
    """4 steps through ExchangeAdam over an RCCL world of one == torch.optim.Adam(fused) on the same gradients — in both exchange
    modes: RCCL's reduce_scatter_tensor and the IN-PLACE all_gather_into_tensor (input = this rank's slice of the output) of `rs_ag`
    are exercised on the device before any multi-GPU run depends on them"""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from nerfacc_amd import sharding
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
torch.manual_seed(0)
a = [torch.nn.Parameter(torch.randn(1 << 20, device=dev)), torch.nn.Parameter(torch.randn(300, 7, device=dev))]
b = [torch.nn.Parameter(x.detach().clone()) for x in a]
oa = torch.optim.Adam(a, lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
ob = sharding.ExchangeAdam(b, lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=4, overlap_backward=True, mode=%r)
ob.timing = True
for it in range(4):
    for ps, o in ((a, oa), (b, ob)):
        o.zero_grad()
        (sum(((p * 1.3 - 0.2) ** 2).sum() for p in ps) * (it + 1)).backward()
        o.step()
st = ob.comm_stats()
assert st["steps"] == 4 and st["window_ms"] > 0, st
for x, y in zip(a, b):
    assert torch.allclose(x, y, atol=1e-6, rtol=1e-5), (x - y).abs().max()
dist.destroy_process_group()
print("ok", st)
''' % (ROOT, str(_free_port()), mode)
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                         env=_clean_env())
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


# Several processes on ONE GPU (what the tests below do to run the N > 1 command on a one-GPU box) is not how the library is deployed
# — one process per device — and this platform was seen to lose part of a kernel's output under it: once in ~60 runs of the
# eight-rank command every eighth workgroup of a count launch (one XCD's share) had left none of its stores in memory, the offsets
# built from the garbage counts were wild and the run died with a GPU memory fault (profiles/r06_oversubscription.md: the captured
# values; no kernel of the library or of torch is exempt, the faults were reported from both).  The library now raises on such totals
# instead of storing through them, and a rank of bench.py that raises ends its job at once (os._exit: no teardown collective the other
# ranks never join).  A run that dies with one of THESE signatures — or has to be killed for saying nothing within its time limit — is
# repeated once, loudly; anything else, or a second failure, fails the test.
_PLATFORM_SIGNATURES = ("HSA_STATUS_ERROR_MEMORY", "Memory access fault", "inconsistent totals", "illegal memory access",
                        "killed with its process group")


def _run_group(cmd, timeout, env):
    """subprocess.run whose command gets a process group of its own: a run that outlives `timeout` is killed WITH its ranks (the
    launcher's children would otherwise stay on the GPU) and comes back with return code -9 and the note in its stderr"""
    import signal
    proc = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = proc.communicate()
        err = (err or "") + f"\n[killed with its process group: no result after {timeout} s]\n"
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def _run_sharing_one_device(cmd_of_port, timeout, env):
    for attempt in (0, 1):
        res = _run_group(cmd_of_port(_free_port()), timeout, env)
        hit = [sig for sig in _PLATFORM_SIGNATURES if sig in res.stderr]
        if res.returncode == 0 or attempt == 1 or not hit:
            return res
        import warnings
        note = "processes sharing one GPU: a run died with %s (profiles/r06_oversubscription.md); repeated once" % hit
        warnings.warn(note)
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "oversubscription_faults.log"), "a") as f:
                f.write(note + "\n" + res.stderr[-2000:] + "\n\n")
    return res


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_eight_ranks_multi_tensor_field_on_one_device(mode):
    """the driver's N = 8 command executed once: eight ranks through torch.distributed.run share cuda:0 and
    exchange through gloo; the field is the feature grid + MLP (seven parameter tensors), so ExchangeAdam's gradient hooks fire in
    autograd's order on a multi-tensor graph and the chunks leave from inside backward — and the line carries the comm figures of
    BOTH exchange modes (aux.exchange_modes)."""
    bench_args = ["--gpus", "8", "--steps", "4", "--warmup", "2", "--windows", "1", "--pretrain", "24", "--pool", "32768", "--aux-steps", "3",
                  "--dist-backend", "gloo", "--all-ranks-on-device0", "--field", "grid+mlp", "--exchange-mode", mode, "--no-other-mode"]
    cmd = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + bench_args
    res = _run_sharing_one_device(cmd, 600, _clean_env())
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]            # stdout carries the JSON line and NOTHING else (no RCCL banner)
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and math.isfinite(out["value"]) and out["value"] > 0 and out["config"]["field"] == "grid+mlp"
    modes = out["aux"]["exchange_modes"]
    assert set(modes) == {"allreduce", "rs_ag"} and all(m["ms_per_step"] > 0 and m["comm_ms_per_step"] >= 0 for m in modes.values())
    assert out["comm_window_ms_per_step"] > 0
