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
