"""Build libcrd.so and the crd_run driver in-tree with hipcc for gfx950 (`make -C crdmodel_amd/csrc`)."""
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(jobs=4, verbose=False):
    cmd = ["make", "-C", _CSRC, "-j", str(jobs)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libcrd.so failed (see output above)")
    return os.path.join(os.path.dirname(_CSRC), "libcrd.so")


if __name__ == "__main__":
    print(build(verbose=True))
