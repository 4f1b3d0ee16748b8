"""Build libnerfacc_hip.so (hipcc, gfx950) and the torch extension _hip*.so in-tree:  python -m nerfacc_amd.build"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, "-j4", "all", "ext"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libnerfacc_hip.so / the torch extension failed (see output above)")
    return os.path.join(os.path.dirname(CSRC), "libnerfacc_hip.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
