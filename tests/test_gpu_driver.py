"""End to end on the GPU: `<program> <ini>` (the reference's command line) -> per-subdomain text files -> the plot
script's loader logic -> compared with the oracle's trajectory at the output times."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import crdmodel_amd as crd
from crdmodel_amd import post
from conftest import GOLDEN, ROOT, rel_err
from oracle import crd_oracle as co
from test_io_formats import load_like_the_plot_script

pytestmark = pytest.mark.gpu
INI = os.path.join(GOLDEN, "ini", "small_run.ini")
BIN = os.path.join(ROOT, "crdmodel_amd", "bin")


def oracle_outputs(cfg):
    p = cfg.params
    g = crd.grid_of(p)
    op = co.make_problem(co.FHN, co.TORUS, g.nx, p.surface_length, p.surface_width, p.diffusion, p.beta, ny=p.ny, t_boundary=p.t_boundary)
    y = co.initial_conditions(op, cfg.wave_length, cfg.wave_width, cfg.wave_inside, 0)
    frames = [y]
    d_tout = cfg.t_final / cfg.output_timestep
    steps = int(np.ceil(d_tout / cfg.dt - 1e-12))
    dt = d_tout / steps
    for k in range(cfg.output_timestep):
        y = co.rk4(op, y, k * d_tout, dt, steps)
        frames.append(y)
    return np.stack(frames)


@pytest.mark.parametrize("argv", [["FHNmodel_torus"], ["crd_run", "--model", "fhn", "--surface", "torus", "--gpus", "2", "--devices", "1", "--quiet"]])
def test_driver_writes_reference_format(gpu_device, tmp_path, argv):
    cfg = crd.load_ini(INI, "fhn", "torus")
    exe = os.path.join(BIN, argv[0])
    r = subprocess.run([exe] + argv[1:] + [INI], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    if "--quiet" not in argv:
        assert "2D FHN model PDE problem on a torus:" in r.stdout and "nx = 16" in r.stdout and "ny = 40" in r.stdout
        assert "100 %" in r.stdout
        # the end-of-run rate line SURVEY section 5 asks for: steps/s, grid-point-steps/s, GB/s under the compulsory-byte model
        m = re.search(r"rate: (\d+) steps in ([0-9.]+) s of stepping = ([0-9.e+]+) steps/s, ([0-9.e+]+) grid-point-steps/s, ([0-9.e+]+) GB/s \(32 B per point-step\)", r.stdout)
        assert m, r.stdout
        steps, secs, sps, pps, gbs = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))
        d_tout = cfg.t_final / cfg.output_timestep
        assert steps == cfg.output_timestep * int(np.ceil(d_tout / cfg.dt - 1e-12)) and secs > 0
        assert pps == pytest.approx(16 * 40 * sps, rel=2e-2) and gbs == pytest.approx(pps * 32 / 1e9, rel=2e-2, abs=0.06)
    want = oracle_outputs(cfg)
    u, meta = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "u")
    v, _ = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "v")  # includeAllVars = 1 in this ini
    assert meta["nprocs"] == (2 if "--gpus" in argv else 1)
    assert u.shape == want[..., 0].shape == (cfg.output_timestep + 1, 40, 16)
    assert np.array_equal(u[0], want[0, ..., 0]) and np.array_equal(v[0], want[0, ..., 1])  # IC row is exact
    assert rel_err(u, want[..., 0]) <= 1e-9 and rel_err(v, want[..., 1]) <= 1e-9
    assert np.abs(u[-1] - u[0]).max() > 0.1  # the wave actually moved
    # the package's own Python-3 loader reads the driver's files the same way
    run = post.load_run(tmp_path, "fhn", "torus", include_all_vars=True)
    assert np.array_equal(run.fields["u"], u) and np.array_equal(run.fields["v"], v) and run.t_final == cfg.t_final


def test_driver_adaptive_mode(gpu_device, tmp_path):
    """`--adaptive`: one error-controlled integration per output interval, as the reference calls ARKode once per output in
    ARK_NORMAL mode (output times do not shorten steps, the rows written are interpolants, the controller's memory carries
    from interval to interval), with ARKode's default explicit pair and controller (CRD_ADAPT_ARKODE) -- checked against the
    oracle's restatement of that published algorithm; `--adaptive-rk43`: the RK4(3) pair of earlier rounds against its own
    restatement."""
    from oracle import arkode_erk as ark

    cfg = crd.load_ini(INI, "fhn", "torus")
    p = cfg.params
    g = crd.grid_of(p)
    op = co.make_problem(co.FHN, co.TORUS, g.nx, p.surface_length, p.surface_width, p.diffusion, p.beta, ny=p.ny, t_boundary=p.t_boundary)
    y0 = co.initial_conditions(op, cfg.wave_length, cfg.wave_width, cfg.wave_inside, 0)
    d_tout = cfg.t_final / cfg.output_timestep
    touts = [cfg.t_final if k + 1 == cfg.output_timestep else (k + 1) * d_tout for k in range(cfg.output_timestep)]

    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--adaptive", INI], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "integrator = ARKode-style ERK on GPU (Zonneveld 5(3)4, PID controller)" in r.stdout and "rtol = 1e-05" in r.stdout and "steps = " in r.stdout
    integ = ark.ArkodeErk(op, 0.0, y0, h_max=crd.stable_dt(p))
    want = np.stack([y0] + [integ.evolve(t)[0] for t in touts])
    u, _ = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "u")
    assert rel_err(u, want[..., 0]) <= 1e-9
    m = re.search(r"steps = (\d+) \(\+(\d+) rejected\)", r.stdout)
    assert m and (int(m.group(1)), int(m.group(2))) == (integ.nst, integ.netf)
    # two slabs: same controller, halos per attempt, norm summed over the slabs -> the same files up to round-off
    out2 = tmp_path / "two"
    out2.mkdir()
    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--adaptive", "--gpus", "2", "--devices", "1", "--quiet", INI],
                       cwd=out2, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    u2, meta2 = load_like_the_plot_script(out2, "FHNmodel_torus", "u")
    assert meta2["nprocs"] == 2 and rel_err(u2, want[..., 0]) <= 1e-9

    # the RK4(3) pair of rounds 1-2 stays available
    out3 = tmp_path / "rk43"
    out3.mkdir()
    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--adaptive-rk43", INI], cwd=out3, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "integrator = adaptive RK4(3) on GPU" in r.stdout
    y, frames, h, dense = y0, [y0], 0.8 * crd.stable_dt(p), {}
    for k, tout in enumerate(touts):
        y, st = co.integrate_adaptive(op, y, k * d_tout, tout, h, h_max=crd.stable_dt(p), dense=dense)
        h = st["h_next"]
        frames.append(y)
    u3, _ = load_like_the_plot_script(out3, "FHNmodel_torus", "u")
    assert rel_err(u3, np.stack(frames)[..., 0]) <= 1e-9


@pytest.mark.parametrize("decomp", [("2x2",), ("mpi",), ("2x2", "--block-contexts"), ("2x2", "--gpus", "1"), ("2x2", "--gpus", "3")])
def test_driver_writes_the_references_np4_block_layout(gpu_device, tmp_path, decomp):
    """`crd_run --gpus 4 --decomp 2x2` (or `--decomp mpi`: MPI_Dims_create of the slab count): the four file sets of the
    reference's `mpirun -np 4` run (util/ShellScripts/runFHNmodelTorus.sh:6) -- subdomain headers with the 2 x 2 extents of
    SetupDecomp, rows of nyl * nxl values per block -- stitched by the plot script's loader logic and compared with the oracle.
    The files' layout is not the computation's: the run steps on phi-slabs (any number of them: 4, 1, 3) and cuts the blocks out
    of their frames; `--block-contexts` computes on the 2 x 2 blocks themselves (staged kernels)."""
    cfg = crd.load_ini(INI, "fhn", "torus")
    argv = ["--gpus", "4", "--devices", "1", "--decomp"] + list(decomp)  # (a later --gpus overrides the first)
    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus"] + argv + [INI], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "nprocs = 4" in r.stdout and "nxl = 8" in r.stdout and "nyl = 20" in r.stdout
    for rank, (c0, c1) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        hdr = open(tmp_path / ("FHNmodel_torus_subdomain.%03d.txt" % rank)).read().split()
        assert tuple(int(v) for v in hdr[:6]) == (16, 40) + crd.block_extents(16, 40, c0, 2, c1, 2)
    assert not (tmp_path / "FHNmodel_torus_subdomain.004.txt").exists()
    want = oracle_outputs(cfg)
    u, meta = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "u")
    v, _ = load_like_the_plot_script(tmp_path, "FHNmodel_torus", "v")
    assert meta["nprocs"] == 4 and u.shape == want[..., 0].shape
    assert np.array_equal(u[0], want[0, ..., 0]) and rel_err(u, want[..., 0]) <= 1e-9 and rel_err(v, want[..., 1]) <= 1e-9


def test_driver_block_layout_with_the_error_controlled_integrator(gpu_device, tmp_path):
    """The block FILE layout no longer limits the integrator: `--decomp 2x2 --adaptive` writes, block by block, the very numbers
    the phi-slab run of the same integrator writes (same slabs underneath: the files are cut out of the same frames).  What
    still needs the slab layout is refused up front: block CONTEXTS with an error-controlled integrator, .npy frames of blocks."""
    run = [os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--gpus", "2", "--devices", "1", "--adaptive", "--quiet"]
    (tmp_path / "slabs").mkdir()
    (tmp_path / "blocks").mkdir()
    r = subprocess.run(run + [INI], cwd=tmp_path / "slabs", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(run + ["--decomp", "2x2", INI], cwd=tmp_path / "blocks", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for name in ("u", "v"):
        a, ma = load_like_the_plot_script(tmp_path / "slabs", "FHNmodel_torus", name)
        b, mb = load_like_the_plot_script(tmp_path / "blocks", "FHNmodel_torus", name)
        assert (ma["nprocs"], mb["nprocs"]) == (2, 4) and np.array_equal(a, b)
        assert np.abs(a[-1] - a[0]).max() > 0.1
    for extra, word in ((["--block-contexts", "--adaptive"], "theta-block contexts"), (["--binary"], "theta-blocks")):
        r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--gpus", "4", "--decomp", "2x2"] + extra + [INI],
                           cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and word in r.stderr, r.stderr


def test_driver_stops_when_any_slab_blows_up(gpu_device, tmp_path):
    """Round-2 advice: the blow-up guard must be sticky over the slabs -- a NaN confined to slab 0 of a two-slab run was
    overwritten by slab 1's finite maximum and the driver kept writing NaN frames.  A step six times the stability limit
    overflows within ~90 steps where the field has structure (the initial rectangle, rows 683-1365: all in slab 0 of two) and
    spreads 4 rows per step, so after the first output interval of 100 steps slab 0 holds NaN and slab 1 is still finite (checked
    with the library directly); the driver must say "Solver failure", stop and exit non-zero as the reference does
    (src/FHNmodel_torus.cpp:424-435)."""
    nx, ny = 64, 4096
    p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
    dt = 6.0 * crd.stable_dt(p)
    cfg = crd.run_config(p, wave_length=1.0 / 6.0, wave_width=0.5, wave_inside=0)
    with crd.LocalGroup(p, 2) as grp:
        grp.upload(crd.initial_conditions(cfg))
        grp.step_rk4(0.0, dt, 100)
        peaks = [s.max_abs() for s in grp.slabs]
    assert not np.isfinite(peaks[0]) and np.isfinite(peaks[1]), peaks  # the premise: only slab 0 has blown up
    ini = tmp_path / "blowup.ini"
    ini.write_text("[Parameters]\ndiffusion = 0.12\nbeta = 1.25\nsurfaceWidth = 20\nsurfaceLength = 80\nwaveLength = %r\nwaveWidth = 0.5\nwaveInside = 0\n"
                   "outputTimestep = 3\ntBoundary = 0\ntFinal = %r\nthetaMesh = %d\nphiMesh = %d\nbetaMin = 0.7\nbetaMax = 1.7\n\n[System]\nincludeAllVars = 0\n"
                   "varyBeta = 0\n[Solver]\ndt = %r\n" % (1.0 / 6.0, 300 * dt, nx, ny, dt))
    r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--gpus", "2", "--devices", "1", "--quiet", str(ini)],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Solver failure, stopping integration" in r.stderr, (r.returncode, r.stderr)
    rows = np.loadtxt(tmp_path / "FHNmodel_torus_u.001.txt", ndmin=2)
    assert rows.shape[0] == 1  # the initial row only: nothing was written after the failing interval


def test_driver_reference_steady_state_option(gpu_device, tmp_path):
    """`GoldbeterModel_torus`-style run with --ref-steady-state: banner and initial row carry the rest state rounded the way the
    reference's print + fscanf pair delivers it (8 decimals), the default run the exact fixed point."""
    ini = os.path.join(GOLDEN, "ini", "goldbeter_shipped.ini")
    rows = {}
    for flag in ("--ref-steady-state", None):
        d = tmp_path / ("ref" if flag else "exact")
        d.mkdir()
        cmd = [os.path.join(BIN, "crd_run"), "--model", "goldbeter", "--surface", "torus"] + ([flag] if flag else []) + [ini]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert ("Stable state values: Z = 0.392, Y = 1.64562" in r.stdout) and "rate: " in r.stdout
        rows[flag] = np.loadtxt(d / "GoldbeterModel_torus_Z.000.txt", ndmin=2)[0]
    zp, _ = crd.steady_state_as_printed("goldbeter", 0.4, 8)
    z, _ = crd.steady_state("goldbeter", 0.4)
    assert set(np.unique(rows["--ref-steady-state"])) == {zp, zp + 1.0} and set(np.unique(rows[None])) == {z, z + 1.0}


def test_config_c1_flat_256_driver(gpu_device, tmp_path):
    """BASELINE config C1: FitzHugh-Nagumo on a 256 x 256 flat periodic grid through the reference's command line
    (`FHNmodel_flat <ini>`, xMesh key as in the shipped data/FHNmodelArgs.ini, surfaceLength = surfaceWidth so ny = nx):
    banner, file set and numbers against the oracle."""
    ini = tmp_path / "c1.ini"
    ini.write_text("[Parameters]\ndiffusion = 0.12\nbeta = 1.25\nsurfaceWidth = 20\t\nsurfaceLength = 20\nwaveLength = 0.1\nwaveWidth = 0.5\n"
                   "outputTimestep = 4\ntBoundary = 0.3\ntFinal = 0.8\nxMesh = 256\nbetaMin = 0.7\nbetaMax = 1.7\n\n[System]\nincludeAllVars = 0\nvaryBeta = 0\n")
    cfg = crd.load_ini(ini, "fhn", "flat")
    g = crd.grid_of(cfg.params)
    assert (g.nx, g.ny) == (256, 256)
    r = subprocess.run([os.path.join(BIN, "FHNmodel_flat"), str(ini)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "2D FHN model PDE problem on a flat surface:" in r.stdout and "Surface length = 20" in r.stdout and "Stable state values: U = -1.25, V = -1.79688" in r.stdout
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("FHNmodel_flat"))
    assert files == ["FHNmodel_flat_subdomain.000.txt", "FHNmodel_flat_u.000.txt", "FHNmodel_flat_v.000.txt"]
    assert os.path.getsize(tmp_path / "FHNmodel_flat_v.000.txt") == 0  # includeAllVars = 0: created, left empty
    p = cfg.params
    op = co.make_problem(co.FHN, co.FLAT, 256, 20.0, 20.0, p.diffusion, p.beta, t_boundary=p.t_boundary)
    y = co.initial_conditions(op, cfg.wave_length, cfg.wave_width, 0, 0)
    d_tout = cfg.t_final / cfg.output_timestep
    steps = int(np.ceil(d_tout / (cfg.dt_safety * crd.stable_dt(p)) - 1e-12))
    frames = [y]
    for k in range(cfg.output_timestep):
        y = co.rk4(op, y, k * d_tout, d_tout / steps, steps, nthreads=4)
        frames.append(y)
    u, meta = load_like_the_plot_script(tmp_path, "FHNmodel_flat", "u")
    assert (meta["nx"], meta["ny"], meta["nt"], meta["xmax"]) == (256, 256, 5, 20.0)
    assert rel_err(u, np.stack(frames)[..., 0]) <= 1e-9


def test_driver_started_by_an_mpi_launcher(gpu_device, tmp_path):
    """Rank 0 of an `mpirun -np 2` start (launcher environment emulated) drives two phi-slabs and writes the two subdomain file
    sets the reference's two ranks would; same numbers as the one-slab run."""
    cfg = crd.load_ini(INI, "fhn", "torus")
    r = subprocess.run([os.path.join(BIN, "FHNmodel_torus"), INI], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PMI_RANK="0", PMI_SIZE="2"))
    assert r.returncode == 0, r.stderr
    assert "started 2 ranks: rank 0 drives 2 phi-slabs" in r.stdout
    run = post.load_run(tmp_path, "fhn", "torus", include_all_vars=True)
    assert run.subdomains.shape == (2, 4)
    want = oracle_outputs(cfg)
    assert rel_err(run.fields["u"], want[..., 0]) <= 1e-9 and rel_err(run.fields["v"], want[..., 1]) <= 1e-9


def test_compiled_arkrhsfn_shim(gpu_device, tmp_path):
    """integration/crd_arkode_shim.c -- the reference's `f(t, y, ydot, user_data)` bound to libcrd -- compiled as C and driven
    through an `ARKRhsFn`-typed pointer by an explicit RK4 loop on host N_Vector arrays (tests/native/shim_selftest.c, with a
    test double for the two SUNDIALS names the shim uses): the trajectory equals the resident-state stepper's to round-off."""
    exe = tmp_path / "shim_selftest"
    lib_dir = os.path.join(ROOT, "crdmodel_amd")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"),
           "-I", os.path.join(ROOT, "tests", "native"), '-DCRD_SHIM_NVECTOR_HEADER="mock_nvector.h"',
           os.path.join(ROOT, "integration", "crd_arkode_shim.c"), os.path.join(ROOT, "tests", "native", "shim_selftest.c"), "-o", str(exe),
           "-L", lib_dir, "-lcrd", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    r = subprocess.run([str(exe), INI], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "through the ARKRhsFn" in r.stdout


def test_driver_binary_side_channel(gpu_device, tmp_path):
    """`crd_run --binary`: <Model>_<surface>_<var>.NNN.npy next to the byte-compatible text files holds the same frames (the
    text's %.16e round-trips doubles exactly, so the two must agree bit for bit); `--binary-only` writes no text rows and the
    Python-3 loader stitches the run from the .npy files alone."""
    from crdmodel_amd import post

    both = tmp_path / "both"
    only = tmp_path / "only"
    for d, flag in ((both, "--binary"), (only, "--binary-only")):
        d.mkdir()
        r = subprocess.run([os.path.join(BIN, "crd_run"), "--model", "fhn", "--surface", "torus", "--gpus", "2", "--devices", "1", flag, "--quiet", INI],
                           cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    u_text, meta = load_like_the_plot_script(both, "FHNmodel_torus", "u")
    for k in range(2):
        frames = np.load(both / ("FHNmodel_torus_u.%03d.npy" % k))
        rows = np.loadtxt(both / ("FHNmodel_torus_u.%03d.txt" % k), ndmin=2)
        assert frames.dtype == np.float64 and np.array_equal(frames.reshape(frames.shape[0], -1), rows)
        assert os.path.getsize(only / ("FHNmodel_torus_u.%03d.txt" % k)) == 0
    run = post.load_run(str(only), "fhn", "torus")
    assert np.array_equal(run.fields["u"], u_text)


def test_seeded_sweep_of_driver_runs(gpu_device, tmp_path):
    """Twelve seeded random runs through the four reference-named executables and `crd_run`: random parameters, mesh keys
    as each program spells them, includeAllVars, varyBeta / justDiffusion, number of GPUs (slabs on this one device), stepper,
    with and without the binary side-channel -- files read back the way the plot scripts read them and compared with the
    oracle's fixed-step trajectory at the output times."""
    rng = np.random.default_rng(1248)
    for case in range(12):
        model = ("fhn", "goldbeter")[int(rng.integers(2))]
        surface = ("torus", "flat")[int(rng.integers(2))]
        nx, ny = int(rng.integers(12, 60)), int(rng.integers(40, 120))
        vary = int(rng.integers(2)) if model == "fhn" else 0
        jd = int(model == "goldbeter" and rng.integers(3) == 0)
        all_vars = int(rng.integers(2))
        beta = float(rng.uniform(0.9, 1.5)) if model == "fhn" else float(rng.uniform(0.2, 0.9))
        L, W, D = float(rng.uniform(40, 100)), float(rng.uniform(10, 25)), float(rng.uniform(0.05, 0.2))
        nt = int(rng.integers(1, 5))
        wl, ww, inside = float(rng.uniform(0.05, 0.3)), float(rng.uniform(0.2, 0.8)), int(rng.integers(2))
        p = crd.make_params(model, surface, nx, L, W, D, beta, ny=ny, vary_beta=vary, beta_min=0.7, beta_max=1.7, just_diffusion=jd)
        dt = float(rng.uniform(0.4, 0.8)) * crd.stable_dt(p)
        t_final = nt * int(rng.integers(5, 16)) * dt
        t_b = float(rng.uniform(0, 1.2)) * t_final
        mesh_key = "thetaMesh" if model == "fhn" else "xMesh"
        d = tmp_path / ("case%02d" % case)
        d.mkdir()
        ini = d / "run.ini"
        ini.write_text("[Parameters]\ndiffusion = %r\nbeta = %r\nsurfaceWidth = %r\nsurfaceLength = %r\nwaveLength = %r\nwaveWidth = %r\nwaveInside = %d\n"
                       "outputTimestep = %d\ntBoundary = %r\ntFinal = %r\n%s = %d\nphiMesh = %d\nbetaMin = 0.7\nbetaMax = 1.7\n\n[System]\nincludeAllVars = %d\n"
                       "varyBeta = %d\njustDiffusion = %d\nicType = 0\n[Solver]\ndt = %r\n" % (D, beta, W, L, wl, ww, inside, nt, t_b, t_final, mesh_key, nx, ny, all_vars,
                                                                                      vary, jd, dt))
        gpus = int(rng.integers(1, 4))
        if ny // gpus < 8:
            gpus = 1
        alias = {("fhn", "torus"): "FHNmodel_torus", ("fhn", "flat"): "FHNmodel_flat", ("goldbeter", "torus"): "GoldbeterModel_torus",
                 ("goldbeter", "flat"): "GoldbeterModel_flat"}[(model, surface)]
        binary = bool(rng.integers(2))
        if gpus == 1 and not binary and rng.integers(2):
            cmd = [os.path.join(BIN, alias), str(ini)]  # the reference's own command line
        else:
            cmd = [os.path.join(BIN, "crd_run"), "--model", model, "--surface", surface, "--gpus", str(gpus), "--devices", "1", "--quiet",
                   "--stepper", ("auto", "staged")[int(rng.integers(2))]] + (["--binary"] if binary else []) + [str(ini)]
            if gpus == 2 and not binary:  # cases 1, 3, 6 -- the reference's own layout for two ranks: MPI_Dims_create(2) = 2 theta-blocks x 1
                cmd = cmd[:-1] + ["--decomp", "mpi", str(ini)]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (case, cmd, r.stderr)
        cfg = crd.load_ini(ini, model, surface)
        pp = cfg.params
        op = co.make_problem(co.FHN if model == "fhn" else co.GOLDBETER, co.TORUS if surface == "torus" else co.FLAT, nx, L, W, D, beta, ny=ny,
                             vary_beta=vary, beta_min=0.7, beta_max=1.7, just_diffusion=jd, t_boundary=pp.t_boundary)
        y = crd.initial_conditions(cfg)
        frames = [y]
        d_tout = cfg.t_final / nt
        steps = int(np.ceil(d_tout / cfg.dt - 1e-12))
        for k in range(nt):
            y = co.rk4(op, y, k * d_tout, d_tout / steps, steps)
            frames.append(y)
        want = np.stack(frames)
        prefix = alias
        names = ("u", "v") if model == "fhn" else ("Z", "Y")
        got0, meta = load_like_the_plot_script(d, prefix, names[0])
        assert meta["nprocs"] == gpus and got0.shape == want[..., 0].shape, (case, meta)
        assert rel_err(got0, want[..., 0]) <= 1e-9, (case, cmd)
        if all_vars:
            got1, _ = load_like_the_plot_script(d, prefix, names[1])
            assert rel_err(got1, want[..., 1]) <= 1e-9 or float(np.max(np.abs(want[..., 1]))) == 0.0, (case, cmd)
        else:
            assert all(os.path.getsize(d / ("%s_%s.%03d.txt" % (prefix, names[1], k))) == 0 for k in range(gpus))
        if binary:
            run = post.load_run(str(d), model, surface, include_all_vars=bool(all_vars))
            assert np.array_equal(run.fields[names[0]], got0)


@pytest.mark.gpu
def test_bench_line_end_to_end(gpu_device):
    """`python bench.py` as the driver runs it (small grid, few steps): exactly one JSON line on stdout with the contract's keys,
    internally consistent (value = points x steps / time, roofline.frac = achieved / peak <= 1, kernel time <= step time), the
    pre-heat reported; then the same through the RCCL self-ring with a pinned launch plan (per_rank diagnostics, halo self-check),
    and the staged stepper alone."""
    import json
    import sys

    def run(*extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "1024", "--steps", "8", "--warmup", "2", "--preheat-ms", "20"] + list(extra),
                           capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    d = run("--cpu-rows", "128", "--cpu-steps", "4", "--staged-steps", "4")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert (d["n_gpus"], d["steps"], d["warmup"], d["unit"], d["dtype"], d["higher_is_better"], d["vs_baseline"]) == (1, 8, 2, "grid-point-steps/s", "f64", True, None)
    assert d["value"] == pytest.approx(1024 * 1024 * 8 / (d["ms_per_step"] * 8e-3), rel=1e-9) and d["config"]["workload"].startswith("fhn_torus_1024x1024")
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu-issue") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "crd_rk4_fused_step_kernel"
    # the same launch on the vector-issue roof, from the build's own kernel table and this run's launch geometry: a fraction of the SIMDs' time,
    # and `bound` names the roof the launch sits closer to (a 1024^2 grid lives in the memory-side cache: issue)
    iss = r["issue"]
    assert 0.05 < r["issue_frac"] <= 1.1 and r["issue_frac"] == pytest.approx(iss["issue_floor_ms"] / r["kernel_ms"]) and iss["valu_instructions_per_trip"] > 100 and iss["simds"] == 1024
    assert (r["bound"] == "valu-issue") == (r["issue_frac"] > 1.05 * r["frac_of_device_streaming"])
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] <= 1.0 and 0.0 < r["kernel_ms"] <= 1.5 * d["ms_per_step"]
    assert r["algorithmic_bytes_per_launch"] == 32 * 1024 * 1024 and r["launches_per_step"] == 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["config"]["preheat"]["steps"] >= 8 and d["staged"]["kernel"].startswith("crd_rk4_stage_kernel") and 0.0 < d["staged"]["frac"] <= 1.0
    assert d["config"]["launch_plan"]["tuned"] == 1  # 1024^2 = 1 Mi points: the smallest launch the library measures a plan for

    d = run("--no-cpu-baseline", "--staged-steps", "0", "--force-rccl", "--launch-plan", "0,1,2,1")
    lp = d["config"]["launch_plan"]
    assert lp["pinned"] and (lp["one_round"], lp["xcd_mapping"], lp["columns_per_lane"], lp["nontemporal_stores"]) == (0, 1, 2, 1) and "cpu_baseline" not in d
    halo = d["config"]["halo"]
    assert halo["transport"] == "rccl" and halo["rccl_comm_count"] == 1 and halo["halo_selfcheck"]["ok"] and halo["slack"]["sweeps"] in (1, 2)
    # (the exchange period is rehearsed on the run's own ring -- 8, 10 or 16 steps: 20 timed steps cross the cycle boundary one to three times)
    assert halo["exchange_period"]["chosen_by"] == "rehearsal" and halo["exchange_period"]["steps"] in (8, 10, 16)
    assert len(d["per_rank"]) == 1 and 1 <= d["per_rank"][0]["exchanges"] <= 3 and d["per_rank"][0]["rows"] == 1024

    d = run("--no-cpu-baseline", "--stepper", "staged")
    assert d["roofline"]["kernel"].startswith("crd_rk4_stage_kernel") and d["roofline"]["launches_per_step"] == 2 and "staged" not in d  # (stages 2 and 3 are the launches of the dominant kernel)
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 80 * 1024 * 1024
