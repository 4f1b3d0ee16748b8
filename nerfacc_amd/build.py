"""Build libnerfacc_hip.so in-tree with hipcc for gfx950:  python -m nerfacc_amd.build"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libnerfacc_hip.so failed (see output above)")
    from .cuda._backend import LIB_PATH

    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
