"""The reference's OWN tests (tests/test_{rendering,scan,grid,pack,pdf}.py) and example renderers
(examples/utils.py), byte-identical copies under build/ref_suite/ (git-ignored; written by
`tools/run_reference_suite.sh prepare` / `__graft_entry__.build()` where /root/reference exists), run
against the `nerfacc` alias of this repository.  Skipped where the copies do not exist."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITE = os.path.join(ROOT, "build", "ref_suite")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(os.path.join(SUITE, "tests")), reason="build/ref_suite not prepared")]


def _env():
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_reference_tests_pass_against_the_alias():
    p = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-p", "no:cacheprovider", "--rootdir", SUITE,
                        "-c", os.devnull], cwd=SUITE, env=_env(), capture_output=True, text=True, timeout=900)
    tail = p.stdout[-3000:] + p.stderr[-1000:]
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "failed" not in p.stdout and "skipped" not in p.stdout, tail
    n = int(p.stdout.split(" passed")[0].split()[-1])
    assert n >= 20, tail      # 6 grid + 1 pack + 3 pdf + 6 rendering + 4 scan


def test_reference_example_renderers_match_this_repos():
    p = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "tools", "ref_examples_check.py")], cwd=ROOT,
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert p.stdout.count("| PASS |") >= 10 and "FAIL" not in p.stdout, p.stdout[-4000:]
