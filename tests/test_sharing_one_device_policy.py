"""The repeat-once policy of the tests that put several processes on one GPU (tests/test_gpu_bench_multirank.py:
_run_sharing_one_device, profiles/r06_oversubscription.md), exercised on the CPU with stand-in commands: a run that dies with one of
the platform's signatures is repeated once; any other failure is not; a second failure stands."""
import os
import sys
import warnings

import test_gpu_bench_multirank as M


def _cmd(tmp_path, script):
    path = os.path.join(tmp_path, "fake.py")
    with open(path, "w") as f:
        f.write(script)
    return lambda port: [sys.executable, path, os.path.join(tmp_path, "state")]


FAIL_ONCE = """
import os, sys
state = sys.argv[1]
n = int(open(state).read()) if os.path.exists(state) else 0
open(state, "w").write(str(n + 1))
if n < %d:
    sys.stderr.write("%s\\n")
    sys.exit(1)
print("ok")
"""


def test_platform_signature_is_repeated_once(tmp_path):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = M._run_sharing_one_device(_cmd(str(tmp_path), FAIL_ONCE % (1, "HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: beyond the largest legal address")), 60, dict(os.environ))
    assert res.returncode == 0 and "ok" in res.stdout
    assert any("repeated once" in str(x.message) for x in w)
    assert open(os.path.join(str(tmp_path), "state")).read() == "2"


def test_second_failure_stands(tmp_path):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = M._run_sharing_one_device(_cmd(str(tmp_path), FAIL_ONCE % (5, "inconsistent totals")), 60, dict(os.environ))
    assert res.returncode == 1
    assert open(os.path.join(str(tmp_path), "state")).read() == "2"


def test_other_failures_are_not_repeated(tmp_path):
    res = M._run_sharing_one_device(_cmd(str(tmp_path), FAIL_ONCE % (1, "AssertionError: replicas differ")), 60, dict(os.environ))
    assert res.returncode == 1
    assert open(os.path.join(str(tmp_path), "state")).read() == "1"


def test_a_silent_run_is_killed_with_its_children(tmp_path):
    script = """
import os, subprocess, sys, time
state = sys.argv[1]
n = int(open(state).read()) if os.path.exists(state) else 0
open(state, "w").write(str(n + 1))
if n == 0:
    child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
    open(state + ".child", "w").write(str(child.pid))
    time.sleep(600)
print("ok")
"""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = M._run_sharing_one_device(_cmd(str(tmp_path), script), 5, dict(os.environ))
    assert res.returncode == 0 and "ok" in res.stdout
    pid = int(open(os.path.join(str(tmp_path), "state.child")).read())
    import time
    time.sleep(0.5)
    try:                      # the grandchild went with its group (a zombie until reaped by init is fine: signal 0 to a dead pid raises)
        os.kill(pid, 0)
        alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
    except (ProcessLookupError, FileNotFoundError):
        alive = False
    assert not alive


def test_a_rank_of_a_multi_rank_bench_fails_fast(tmp_path):
    """bench.py's main(): under WORLD_SIZE > 1 an exception out of run() ends the process at once with code 1 and the traceback on
    stderr (no interpreter teardown, no process-group destructor waiting for ranks that will never come); a single process re-raises"""
    import subprocess
    code = ("import os, sys; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench\n"
            "def boom(args): raise RuntimeError('nerfacc_amd: sample_occgrid read back inconsistent totals (stand-in)')\n"
            "bench.run = boom\n"
            "import atexit; atexit.register(lambda: print('ATEXIT RAN', file=sys.stderr))\n"
            "bench.main()\n") % M.ROOT
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    res = subprocess.run([sys.executable, "-c", code], cwd=M.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert res.returncode == 1 and "inconsistent totals" in res.stderr and "ATEXIT RAN" not in res.stderr, res.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
    res = subprocess.run([sys.executable, "-c", code], cwd=M.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert res.returncode == 1 and "inconsistent totals" in res.stderr and "ATEXIT RAN" in res.stderr, res.stderr[-2000:]
