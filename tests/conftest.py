"""Shared fixtures.  `-m gpu` tests need an MI355X and call the HIP path through the C ABI; everything else runs on CPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A checkout without the built library (the .so files are not tracked) gets it built here, in-tree, by the same recipe as
    __graft_entry__.build(); an existing build is left alone."""
    pkg = os.path.join(ROOT, "crdmodel_amd")
    if not (os.path.exists(os.path.join(pkg, "libcrd.so")) and os.path.exists(os.path.join(pkg, "bin", "crd_run"))):
        from crdmodel_amd.build import build

        build()


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(d["meta"]))
    return meta, {k: d[k] for k in d.files if k != "meta"}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def oracle_problem(meta):
    """crd_oracle Problem for a golden case's metadata."""
    from oracle import crd_oracle as co

    return co.make_problem({"fhn": co.FHN, "goldbeter": co.GOLDBETER}[meta["model"]], {"torus": co.TORUS, "flat": co.FLAT}[meta["surface"]],
                           meta["nx"], meta["surface_length"], meta["surface_width"], meta["diffusion"], meta["beta"],
                           ny=meta["ny_override"], beta_min=meta["beta_min"], beta_max=meta["beta_max"], vary_beta=meta["vary_beta"],
                           just_diffusion=meta["just_diffusion"], t_boundary=meta["t_boundary"])


def crd_params(meta, precision="f64"):
    """libcrd Params for a golden case's metadata."""
    import crdmodel_amd as crd

    return crd.make_params(meta["model"], meta["surface"], meta["nx"], meta["surface_length"], meta["surface_width"], meta["diffusion"],
                           meta["beta"], ny=meta["ny_override"], beta_min=meta["beta_min"], beta_max=meta["beta_max"],
                           vary_beta=meta["vary_beta"], just_diffusion=meta["just_diffusion"], t_boundary=meta["t_boundary"],
                           precision=precision)


def rel_err(a, b):
    """max-norm error scaled by max |b| ('relative, max-norm, scaled by max|ydot|')."""
    scale = float(np.max(np.abs(b)))
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b))) / (scale if scale > 0 else 1.0)


@pytest.fixture(scope="session")
def gpu_device():
    """Fail loudly (not skip) when a -m gpu test runs without the HIP library or a device."""
    import crdmodel_amd as crd

    crd._capi.lib()
    return 0
