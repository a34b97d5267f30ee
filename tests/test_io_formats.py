"""File-format contract with the reference's Python utilities: the writer's bytes, a Python-3 restatement of the loader
logic of util/FHNmodel/plot_FHNmodel_torus.py:26-87, and the command-line surface of the driver."""
import os
import subprocess

import numpy as np
import pytest

import crdmodel_amd as crd
from conftest import GOLDEN, ROOT

INI = os.path.join(GOLDEN, "ini")
BIN = os.path.join(ROOT, "crdmodel_amd", "bin")


def load_like_the_plot_script(directory, prefix, var):
    """Python-3 restatement of the reference loader: probe subdomain files until one is missing, np.loadtxt the header
    (nx ny is ie js je xmin xmax tfinal) and the data rows, reshape each row to (nyl, nxl) and paste it at [js:je+1, is:ie+1]."""
    nprocs = 0
    while os.path.exists(os.path.join(directory, "%s_subdomain.%03d.txt" % (prefix, nprocs))):
        nprocs += 1
    assert nprocs >= 1
    results, meta = None, None
    for i in range(nprocs):
        subd = np.loadtxt(os.path.join(directory, "%s_subdomain.%03d.txt" % (prefix, i)), dtype=np.float64)
        nx, ny = int(subd[0]), int(subd[1])
        istart, iend, jstart, jend = (int(v) for v in subd[2:6])
        data = np.loadtxt(os.path.join(directory, "%s_%s.%03d.txt" % (prefix, var, i)), dtype=np.double, ndmin=2)
        nt = data.shape[0]
        if results is None:
            results = np.zeros((nt, ny, nx))
            meta = dict(nx=nx, ny=ny, xmin=subd[6], xmax=subd[7], tfinal=subd[8], nprocs=nprocs, nt=nt)
        assert data.shape == (nt, (jend - jstart + 1) * (iend - istart + 1))
        for t in range(nt):
            results[t, jstart:jend + 1, istart:iend + 1] = data[t].reshape(jend - jstart + 1, iend - istart + 1)
    return results, meta


@pytest.mark.parametrize("n_slabs", [1, 3])
def test_writer_round_trip(tmp_path, n_slabs):
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    g = crd.grid_of(cfg.params)
    rng = np.random.default_rng(3)
    frames = [rng.standard_normal((g.ny, g.nx, 2)) * 10.0 ** rng.integers(-8, 8) for _ in range(3)]
    frames[1][0, 0, 0], frames[1][0, 1, 0], frames[1][0, 2, 1] = 0.0, -0.0, 5e-324
    for k in range(n_slabs):
        js, je = crd.slab_extents(g.ny, k, n_slabs)
        with crd.Writer(cfg, tmp_path, k, n_slabs) as w:
            for f in frames:
                w.write_row(f[js:je + 1])
    u, meta = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "u")
    v, _ = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "v")
    assert meta["nprocs"] == n_slabs and (meta["nx"], meta["ny"], meta["nt"]) == (g.nx, g.ny, 3)
    assert meta["tfinal"] == cfg.t_final and meta["xmin"] == 0.0 and meta["xmax"] == pytest.approx(2 * np.pi, abs=1e-6)
    for t, f in enumerate(frames):
        assert np.array_equal(u[t], f[..., 0]) and np.array_equal(v[t], f[..., 1])  # %.16e round-trips doubles exactly


def test_writer_bytes(tmp_path):
    """Header "%li  %li  %li  %li  %li  %li %f %f %f\\n" and rows of " %.16e" (src/FHNmodel_torus.cpp:379-380,397-405)."""
    p = crd.make_params("goldbeter", "flat", 3, 40.0, 20.0, 0.12, 0.4)
    cfg = crd.run_config(p, t_final=4.0, include_all_vars=0)
    y = np.array([[[1.0, 9.0], [-2.5, 9.0], [1e-300, 9.0]]] * 6)
    with crd.Writer(cfg, tmp_path) as w:
        w.write_row(y)
    assert open(tmp_path / "GoldbeterModel_flat_subdomain.000.txt").read() == "3  6  0  2  0  5 0.000000 20.000000 4.000000\n"
    row = " 1.0000000000000000e+00 -2.5000000000000000e+00 1.0000000000000000e-300" * 6 + "\n"
    assert row == "".join(" %.16e" % v for v in y[..., 0].ravel()) + "\n"  # C and Python format doubles identically
    assert open(tmp_path / "GoldbeterModel_flat_Z.000.txt").read() == row
    # the second-variable file is always created, and stays empty unless includeAllVars == 1 (:388-389,399-402)
    assert open(tmp_path / "GoldbeterModel_flat_Y.000.txt").read() == ""


def test_driver_command_line_surface(tmp_path):
    """`<program>` with the wrong argument count prints the reference's usage line and exits 1 (src/FHNmodel_torus.cpp:151-155);
    an unusable ini file is reported before any GPU work starts."""
    exe = os.path.join(BIN, "FHNmodel_torus")
    assert os.path.islink(exe) or os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage: ") and r.stderr.endswith("<Config file path>")
    r = subprocess.run([exe, "a.ini", "b.ini"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage: " in r.stderr
    r = subprocess.run([exe, str(tmp_path / "missing.ini")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    r = subprocess.run([exe, os.path.join(INI, "temp_shipped.ini")], capture_output=True, text=True)
    assert r.returncode == 1 and "betaMin" in r.stderr
    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", os.path.join(INI, "small_run.ini")], capture_output=True, text=True)
    assert r.returncode == 1 and "--surface" in r.stderr


def test_driver_under_an_mpi_launcher_leaves_the_work_to_rank_zero(tmp_path):
    """`mpirun -np N <exe> <ini>` is how the reference is started (util/ShellScripts/runFHNmodelTorus.sh:6).  This driver is
    one process for all GPUs: every rank but 0 exits at once, successfully and without touching the output directory."""
    for env in (dict(PMI_RANK="1", PMI_SIZE="2"), dict(OMPI_COMM_WORLD_RANK="3", OMPI_COMM_WORLD_SIZE="4")):
        r = subprocess.run([os.path.join(BIN, "FHNmodel_torus"), os.path.join(INI, "small_run.ini")], cwd=tmp_path, capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=60)
        assert r.returncode == 0 and r.stdout == "" and r.stderr == ""
        assert os.listdir(tmp_path) == []


@pytest.mark.parametrize("value_bytes", [8, 4])
def test_npy_side_channel_round_trip(tmp_path, value_bytes):
    """crd_npy_writer: one .npy per slab and variable holding the text file's frames; numpy reads it back exactly, the frame
    count in the header is the number of frames actually appended, and the post-processing loader prefers it over the text."""
    import ctypes as C

    L = crd._capi.lib()
    p = crd.make_params("goldbeter", "flat", 12, 20.0, 20.0, 0.12, 0.4, precision="f64")
    cfg = crd.run_config(p, output_timestep=5, t_final=1.0)
    g = crd.grid_of(p)
    dtype = np.float64 if value_bytes == 8 else np.float32
    rng = np.random.default_rng(5)
    for slab in range(2):
        js, je = crd.slab_extents(g.ny, slab, 2)
        frames = rng.standard_normal((3, je - js + 1, g.nx)).astype(dtype)
        for var, name in ((0, "Z"), (1, "Y")):
            h = C.c_void_p()
            assert L.crd_npy_writer_open(C.byref(cfg), str(tmp_path).encode(), slab, 2, var, value_bytes, C.byref(h)) == 0
            for f in frames:
                f = np.ascontiguousarray(f + var)
                assert L.crd_npy_writer_append(h, f.ctypes.data) == 0
            assert L.crd_npy_writer_close(h) == 0
            back = np.load(tmp_path / ("GoldbeterModel_flat_%s.%03d.npy" % (name, slab)))
            assert back.dtype == dtype and back.shape == frames.shape and np.array_equal(back, frames + var)
    h = C.c_void_p()
    assert L.crd_npy_writer_open(C.byref(cfg), str(tmp_path / "nope").encode(), 0, 1, 0, 8, C.byref(h)) != 0 and not h
    assert L.crd_npy_writer_open(C.byref(cfg), str(tmp_path).encode(), 0, 1, 2, 8, C.byref(h)) != 0
    assert L.crd_npy_writer_open(C.byref(cfg), str(tmp_path).encode(), 0, 1, 0, 3, C.byref(h)) != 0
