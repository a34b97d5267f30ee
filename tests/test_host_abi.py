"""CPU-side checks of libcrd's C ABI: the library loads, exports every symbol include/crd.h declares, and its host-side
helpers (geometry, slabs, stable states, initial conditions, ini reader) agree with the oracle.  No kernel is launched."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import crdmodel_amd as crd
from conftest import GOLDEN, ROOT
from oracle import crd_oracle as co

INI = os.path.join(GOLDEN, "ini")


def declared_functions():
    text = open(os.path.join(ROOT, "include", "crd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 30
    L = C.CDLL(crd._capi.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # and the ctypes table binds exactly the declared set, so a header change cannot go unbound
    assert sorted(crd._capi._SIGNATURES) == names
    header_version = int(re.search(r"#define CRD_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "crd.h")).read()).group(1))
    assert crd._capi.lib().crd_abi_version() == crd._capi.ABI_VERSION == header_version


def test_status_strings():
    L = crd._capi.lib()
    assert L.crd_status_string(0) == b"ok"
    assert len({L.crd_status_string(-k) for k in range(0, 8)}) == 8


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device crd_create fails with CRD_EHIP; there is no CPU path to fall back to."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    p = crd.make_params("fhn", "torus", 32, 80.0, 20.0, 0.12, 1.25)
    with pytest.raises(crd._capi.CrdError) as e:
        crd.Slab(p)
    assert e.value.status == crd._capi.EHIP and "no CPU fallback" in str(e.value)


def test_geometry_matches_oracle_and_golden():
    geo = json.load(open(os.path.join(GOLDEN, "geometry.json")))
    for e in geo["ny"]:
        p = crd.make_params("fhn", e["surface"], e["nx"], e["L"], e["W"], 0.12, 1.25)
        g = crd.grid_of(p)
        assert (g.nx, g.ny, g.dx, g.dy, g.R, g.r) == (e["nx"], e["ny"], e["dx"], e["dy"], e["R"], e["r"])
    # phiMesh override keeps everything else
    g = crd.grid_of(crd.make_params("fhn", "torus", 8192, 80.0, 20.0, 0.12, 1.25, ny=8192))
    assert (g.nx, g.ny) == (8192, 8192) and g.dy == (2.0 * 3.1415926535897932) / 8191.0


def test_bad_params_are_rejected():
    L = crd._capi.lib()
    g = crd._capi.Grid()
    for kw in (dict(nx=1), dict(surface_width=0.0), dict(diffusion=float("nan")), dict(model=7), dict(precision=3)):
        p = crd.make_params("fhn", "torus", 32, 80.0, 20.0, 0.12, 1.25)
        for k, v in kw.items():
            setattr(p, k, v)
        assert L.crd_grid_from_params(C.byref(p), C.byref(g)) == crd._capi.EINVAL
    # horn torus (R = r) needs an explicit phiMesh
    p = crd.make_params("fhn", "torus", 32, 20.0, 20.0, 0.12, 1.25)
    assert L.crd_grid_from_params(C.byref(p), C.byref(g)) == crd._capi.EINVAL


def test_slab_extents_follow_setupdecomp():
    geo = json.load(open(os.path.join(GOLDEN, "geometry.json")))
    for e in geo["slabs"]:
        got = [list(crd.slab_extents(e["ny"], k, e["n_slabs"])) for k in range(e["n_slabs"])]
        assert got == e["extents"]
        assert got[0][0] == 0 and got[-1][1] == e["ny"] - 1
        assert all(got[k][1] + 1 == got[k + 1][0] for k in range(len(got) - 1))  # contiguous cover
    with pytest.raises(crd._capi.CrdError):
        crd.slab_extents(100, 4, 4)


def test_steady_states():
    geo = json.load(open(os.path.join(GOLDEN, "geometry.json")))
    for e in geo["steady"]:
        got = crd.steady_state(e["model"], e["beta"])
        assert got == pytest.approx(tuple(e["state"]), rel=1e-14)
    assert crd.steady_state("fhn", 1.25) == (-1.25, 1.25 ** 3 - 3 * 1.25)
    # beta = 0.4 -> (0.392, 1.64562146714406)
    assert crd.steady_state("goldbeter", 0.4) == pytest.approx((0.392, 1.64562146714406), rel=1e-14)


def test_goldbeter_steady_state_as_the_reference_reads_it(tmp_path):
    """`[Solver] steadyStateDigits = 8` / `crd_run --ref-steady-state`: the Goldbeter rest state the way the reference's programs
    receive it -- numpy's print of one-element arrays, 8 digits behind the decimal point (util/GoldbeterModel/
    SolveGoldbeterODE.py:111: `print Z[-1], Y[-1]` -> "[ 0.392] [ 1.64562147]"), read back by fscanf("[%lf] [%lf]")
    (src/GoldbeterModel_torus.cpp:254-261) -- at beta = 0.4 (inside the oscillatory window) and 0.14 (outside)."""
    for beta in (0.4, 0.14):
        z, y = crd.steady_state("goldbeter", beta)
        zp, yp = crd.steady_state_as_printed("goldbeter", beta, 8)
        # exactly what Python's own "%.8f" -> float round trip gives (numpy then drops trailing zeros: same number)
        assert (zp, yp) == (float("%.8f" % z), float("%.8f" % y))
        assert abs(zp - z) <= 5e-9 and 0 < abs(yp - y) <= 5e-9
        text = "[ %s] [ %s]" % (np.format_float_positional(zp, precision=8, trim="-"), np.format_float_positional(yp, precision=8, trim="-"))
        back = [float(t.strip("[] ")) for t in text.split("] [")]  # what fscanf("[%lf] [%lf]") extracts
        assert back == [zp, yp]
    assert crd.steady_state_as_printed("goldbeter", 0.4, 8) == (0.392, 1.64562147)
    assert crd.steady_state_as_printed("goldbeter", 0.4, 0) == crd.steady_state("goldbeter", 0.4)
    assert crd.steady_state_as_printed("fhn", 1.3, 8) == crd.steady_state("fhn", 1.3)  # the FHN state is computed in C++ (:242-244)
    # through the ini key into the initial conditions: the rectangle sits on (Zs + 1, Ys + 1) of the rounded state
    base = open(os.path.join(INI, "goldbeter_shipped.ini")).read()
    ini = tmp_path / "gb.ini"
    ini.write_text(base + "\n[Solver]\nsteadyStateDigits = 8\n")
    cfg = crd.load_ini(ini, "goldbeter", "torus")
    assert cfg.steady_state_decimals == 8 and crd.load_ini(os.path.join(INI, "goldbeter_shipped.ini"), "goldbeter", "torus").steady_state_decimals == 0
    y0 = crd.initial_conditions(cfg)
    zp, yp = crd.steady_state_as_printed("goldbeter", cfg.params.beta, 8)
    assert set(np.unique(y0[..., 0])) == {zp, zp + 1.0} and set(np.unique(y0[..., 1])) == {yp, yp + 1.0}
    exact = crd.initial_conditions(crd.load_ini(os.path.join(INI, "goldbeter_shipped.ini"), "goldbeter", "torus"))
    assert 0 < np.abs(exact - y0).max() <= 5e-9
    bad = tmp_path / "bad.ini"
    bad.write_text(base + "\n[Solver]\nsteadyStateDigits = 40\n")
    with pytest.raises(crd._capi.CrdError):
        crd.load_ini(bad, "goldbeter", "torus")
    # [Solver] exchangePeriod (round 4): 0 / absent = the driver chooses, 3 .. 16 = crd_set_exchange_period for every slab of the run
    assert cfg.exchange_period == 0
    ini.write_text(base + "\n[Solver]\nexchangePeriod = 12\n")
    assert crd.load_ini(ini, "goldbeter", "torus").exchange_period == 12
    for v in (2, 17, -1):
        bad.write_text(base + "\n[Solver]\nexchangePeriod = %d\n" % v)
        with pytest.raises(crd._capi.CrdError):
            crd.load_ini(bad, "goldbeter", "torus")


IC_CASES = [
    ("fhn", "torus", dict(wave_inside=0), {}),
    ("fhn", "torus", dict(wave_inside=1), {}),
    ("fhn", "torus", dict(wave_inside=0), dict(vary_beta=1, beta_min=0.7, beta_max=1.7)),
    ("fhn", "flat", {}, {}),
    ("fhn", "flat", {}, dict(vary_beta=1, beta_min=0.7, beta_max=1.7)),
    ("goldbeter", "torus", dict(wave_inside=1, wave_length=0.2), {}),
    ("goldbeter", "torus", dict(wave_inside=0, wave_length=0.2), {}),
    ("goldbeter", "flat", dict(wave_length=0.2), {}),
    ("goldbeter", "flat", dict(wave_length=0.2, ic_type=0), dict(vary_beta=1, beta_max=1.0)),
    ("goldbeter", "flat", dict(wave_length=0.2, ic_type=1), dict(vary_beta=1, beta_max=1.0)),
    ("goldbeter", "flat", dict(wave_length=0.2, ic_type=2), dict(vary_beta=1, beta_max=1.0)),
    ("goldbeter", "torus", dict(wave_inside=0, wave_length=0.2, ic_type=1), dict(vary_beta=1)),
]


@pytest.mark.parametrize("model,surface,ickw,pkw", IC_CASES)
def test_initial_conditions_match_oracle(model, surface, ickw, pkw):
    """The four programs' IC rules (rectangle inside / outside, Goldbeter icType 0/1/2 incl. the unseeded rand())."""
    beta = 1.25 if model == "fhn" else 0.4
    p = crd.make_params(model, surface, 40, 80.0, 20.0, 0.12, beta, **pkw)
    kw = dict(wave_length=0.1, wave_width=0.5)
    kw.update(ickw)
    cfg = crd.run_config(p, **kw)
    y = crd.initial_conditions(cfg)
    op = co.make_problem({"fhn": co.FHN, "goldbeter": co.GOLDBETER}[model], {"torus": co.TORUS, "flat": co.FLAT}[surface], 40, 80.0, 20.0,
                         0.12, beta, **pkw)
    ref = co.initial_conditions(op, kw["wave_length"], kw["wave_width"], kw.get("wave_inside", 0), kw.get("ic_type", 0),
                                crd.steady_state(model, beta))
    assert y.shape == ref.shape == (op.ny, op.nx, 2)
    assert np.array_equal(y, ref)
    if not pkw.get("vary_beta") or kw.get("ic_type") == 1:
        frac = np.mean(y[..., 0] != y[0, 0, 0]) if not (surface == "torus" and kw.get("wave_inside", 0) == 0 and kw.get("ic_type") == 1) else 0.05
        assert 0.0 < frac < 0.2  # a perturbed rectangle exists and is small
    # slab-wise generation == rows of the whole-domain field (except rand(), which restarts per slab like per MPI rank)
    if kw.get("ic_type") != 2:
        js, je = crd.slab_extents(op.ny, 1, 3)
        assert np.array_equal(crd.initial_conditions(cfg, js, je), ref[js:je + 1])


def test_ini_shipped_parameter_sets():
    fhn = crd.load_ini(os.path.join(INI, "fhn_shipped.ini"), "fhn", "torus")  # carries xMesh: accepted for the FHN programs too
    p = fhn.params
    assert (p.nx, p.diffusion, p.beta, p.surface_width, p.surface_length) == (400, 0.12, 1.25, 20.0, 80.0)
    assert (p.vary_beta, p.beta_min, p.beta_max, p.t_boundary) == (1, 0.7, 1.7, 38.0)
    assert (fhn.wave_length, fhn.wave_width, fhn.wave_inside, fhn.output_timestep, fhn.t_final, fhn.include_all_vars) == (0.1, 0.5, 0, 20, 50.0, 0)
    assert crd.grid_of(p).ny == 1600
    assert (fhn.dt, fhn.n_gpus, fhn.stepper, p.precision) == (0.0, 1, 0, 0)

    gb_t = crd.load_ini(os.path.join(INI, "goldbeter_shipped.ini"), "goldbeter", "torus")
    gb_f = crd.load_ini(os.path.join(INI, "goldbeter_shipped.ini"), "goldbeter", "flat")
    assert (gb_t.params.nx, gb_t.params.beta, gb_t.wave_inside, gb_t.output_timestep) == (100, 0.4, 1, 5)
    # Goldbeter torus never reads betaMin / betaMax / icType (they stay 0); Goldbeter flat does
    assert (gb_t.params.beta_max, gb_t.ic_type) == (0.0, 0) and (gb_f.params.beta_max, gb_f.ic_type) == (1.0, 2)
    assert crd.grid_of(gb_f.params).ny == 400


def test_ini_missing_and_malformed_keys(tmp_path):
    # data/temp.ini has thetaMesh but no betaMin/betaMax while varyBeta = 1: the FHN programs abort on it, so do we
    with pytest.raises(crd._capi.CrdError) as e:
        crd.load_ini(os.path.join(INI, "temp_shipped.ini"), "fhn", "torus")
    assert e.value.status == crd._capi.EPARSE and "betaMin" in str(e.value)
    t = crd.load_ini(os.path.join(INI, "temp_shipped.ini"), "goldbeter", "torus")  # reads thetaMesh in place of xMesh
    assert t.params.nx == 200 and t.params.vary_beta == 1

    good = open(os.path.join(INI, "small_run.ini")).read()
    cases = {
        "no_key": (good.replace("diffusion = 0.12\n", ""), crd._capi.EPARSE, "diffusion"),
        "not_a_number": (good.replace("beta = 1.25", "beta = 1.25x"), crd._capi.EPARSE, "conversion"),
        "duplicate": (good.replace("beta = 1.25\n", "beta = 1.25\nbeta = 2\n"), crd._capi.EPARSE, "duplicate"),
        "no_equals": (good.replace("beta = 1.25", "beta 1.25"), crd._capi.EPARSE, "'='"),
        "bad_section": (good.replace("[System]", "[System"), crd._capi.EPARSE, "unmatched"),
        "no_system": (good.replace("varyBeta = 0\n", ""), crd._capi.EPARSE, "varyBeta"),
        "bad_value": (good.replace("thetaMesh = 16", "thetaMesh = 1"), crd._capi.EINVAL, "nx"),
    }
    for name, (text, status, needle) in cases.items():
        f = tmp_path / (name + ".ini")
        f.write_text(text)
        with pytest.raises(crd._capi.CrdError) as e:
            crd.load_ini(f, "fhn", "torus")
        assert e.value.status == status and needle in str(e.value), (name, str(e.value))
    with pytest.raises(crd._capi.CrdError) as e:
        crd.load_ini(tmp_path / "absent.ini", "fhn", "torus")
    assert e.value.status == crd._capi.EIO

    small = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    assert (small.params.nx, small.params.ny, small.dt, small.include_all_vars) == (16, 40, 0.02, 1)


def test_stable_dt_bounds_the_diffusion_limit():
    """SURVEY 8d quotes 1.25e-4 (4096^2) and 3.1e-5 (8192^2) as RK4 limits of the diffusion operator alone."""
    for n, lim in ((4096, 1.25e-4), (8192, 3.1e-5)):
        dt = crd.stable_dt(crd.make_params("fhn", "torus", n, 80.0, 20.0, 0.12, 1.25, ny=n))
        assert 0.8 * lim < dt <= 1.02 * lim


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/crd.h must be consumable by a C compiler (the reference's FFI would be cgo / ctypes / a C++ TU): compile a
    C99 program against it with gcc -pedantic, link libcrd.so, and run the host-side entry points."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "crd.h"
int main(void) {
	crd_params p; crd_grid g; int64_t js, je; double s0, s1; crd_halo_op ops[4]; crd_adaptive_options ao;
	memset(&p, 0, sizeof p);
	p.model = CRD_MODEL_FHN; p.surface = CRD_SURFACE_TORUS; p.nx = 100; p.surface_length = 100.0; p.surface_width = 20.0;
	p.diffusion = 0.12; p.beta = 1.25; p.precision = CRD_PRECISION_F64;
	if (crd_abi_version() != CRD_ABI_VERSION) return 1;
	if (crd_grid_from_params(&p, &g) != CRD_OK || g.ny != 499) return 2;
	if (crd_slab_extents(g.ny, 3, 4, &js, &je) != CRD_OK || je != 498) return 3;
	if (crd_steady_state(CRD_MODEL_FHN, 1.25, &s0, &s1) != CRD_OK || s0 != -1.25) return 4;
	if (crd_halo_plan(0, 2, 64, 16, ops) != CRD_OK || ops[0].peer != 1 || ops[2].row_begin != -16) return 5;
	if (crd_adaptive_defaults(&ao) != CRD_OK || ao.rtol != 1e-5) return 6;
	if (!(crd_stable_dt(&p) > 0.0)) return 7;
	printf("%s %ld\n", crd_status_string(CRD_EPARSE), (long)sizeof(crd_run_config));
	printf("layout %ld %ld %ld %ld %ld %ld %ld %ld %ld %ld\n", (long)sizeof(crd_params), (long)sizeof(crd_grid), (long)sizeof(crd_adaptive_options),
	       (long)sizeof(crd_adaptive_stats), (long)sizeof(crd_launch_plan), (long)sizeof(crd_step_timing), (long)offsetof(crd_launch_plan, nontemporal_stores),
	       (long)offsetof(crd_launch_plan, ms_default), (long)offsetof(crd_step_timing, halo_waits), (long)offsetof(crd_adaptive_options, method));
	return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "crdmodel_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lcrd",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert r.stdout.splitlines()[0].split()[:3] == ["ini", "parse", "error"]
    assert int(r.stdout.splitlines()[0].split()[-1]) == C.sizeof(crd._capi.RunConfig)  # the ctypes mirror has the C layout
    k = crd._capi
    layout = [int(v) for v in r.stdout.splitlines()[1].split()[1:]]
    assert layout == [C.sizeof(k.Params), C.sizeof(k.Grid), C.sizeof(k.AdaptiveOptions), C.sizeof(k.AdaptiveStats), C.sizeof(k.LaunchPlan), C.sizeof(k.StepTiming),
                      k.LaunchPlan.nontemporal_stores.offset, k.LaunchPlan.ms_default.offset, k.StepTiming.halo_waits.offset, k.AdaptiveOptions.method.offset]


def test_arkrhsfn_shim_compiles_as_c(tmp_path):
    """integration/crd_arkode_shim.c is plain C against include/crd.h: it compiles (with the test double for the two SUNDIALS
    names it uses) and links against libcrd; running it needs a GPU (tests/test_gpu_driver.py)."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    lib_dir = os.path.join(ROOT, "crdmodel_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"),
           "-I", os.path.join(ROOT, "tests", "native"), '-DCRD_SHIM_NVECTOR_HEADER="mock_nvector.h"',
           os.path.join(ROOT, "integration", "crd_arkode_shim.c"), os.path.join(ROOT, "tests", "native", "shim_selftest.c"),
           "-o", str(tmp_path / "shim"), "-L", lib_dir, "-lcrd", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_arkrhsfn_shim_never_skips_its_broadcast(tmp_path):
    """crd_arkode_attach with nprocs > 1: every rank takes part in the ONE broadcast whatever happened to it before (round-2
    advice: a rank that returned early left the others blocked in bcast), and rank 0's failure reaches every rank through the
    status byte.  Runs without a GPU: there crd_create fails on every rank, which is exactly the situation to get right."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "bcast.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "crd_arkode_shim.h"
static int calls, bytes_seen; static unsigned char root_status;
static int fake_bcast(void *buf, int bytes, void *comm) {
	(void)comm; calls++; bytes_seen = bytes;
	if (root_status) ((unsigned char *)buf)[0] = root_status; /* what a failing rank 0 would have sent */
	return 0;
}
int main(void) {
	crd_run_config cfg; crd_ctx *ctx = (crd_ctx *)1; int rc;
	memset(&cfg, 0, sizeof cfg);
	cfg.params.model = CRD_MODEL_FHN; cfg.params.surface = CRD_SURFACE_TORUS; cfg.params.nx = 32; cfg.params.ny = 64;
	cfg.params.surface_length = 80.0; cfg.params.surface_width = 20.0; cfg.params.diffusion = 0.12; cfg.params.beta = 1.25;
	int have_gpu = crd_device_count() > 0;
	/* rank 0 of 2 */
	calls = 0; root_status = 0;
	if (!have_gpu) { /* (with a GPU this rank would go on to wait for a second rank that does not exist) */
		rc = crd_arkode_attach(&cfg, 0, 2, 0, fake_bcast, NULL, &ctx);
		if (calls != 1 || bytes_seen != 129 || rc != CRD_EHIP || ctx != NULL) { printf("rank0 %d %d %d\n", calls, bytes_seen, rc); return 1; }
	}
	/* rank 1 of 2 told by the status byte that rank 0 failed: one broadcast, an error, no hang in crd_comm_init_rccl */
	calls = 0; root_status = 3; ctx = (crd_ctx *)1;
	rc = crd_arkode_attach(&cfg, 1, 2, 0, fake_bcast, NULL, &ctx);
	if (calls != 1 || rc == CRD_OK || ctx != NULL) { printf("rank1 %d %d\n", calls, rc); return 2; }
	/* no broadcast function at all */
	rc = crd_arkode_attach(&cfg, 1, 2, 0, NULL, NULL, &ctx);
	if (rc != CRD_EINVAL) return 3;
	printf("ok\n");
	return 0;
}
''')
    lib_dir = os.path.join(ROOT, "crdmodel_amd")
    exe = tmp_path / "bcast"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"),
           "-I", os.path.join(ROOT, "tests", "native"), '-DCRD_SHIM_NVECTOR_HEADER="mock_nvector.h"',
           os.path.join(ROOT, "integration", "crd_arkode_shim.c"), str(src), "-o", str(exe), "-L", lib_dir, "-lcrd", "-Wl,-rpath," + lib_dir,
           "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_block_decomposition_follows_setupdecomp(tmp_path):
    """The reference's 2-D layout: MPI_Dims_create(np, 2) as SetupDecomp calls it
    (src/FHNmodel_torus.cpp:724-728), block extents (:750-753) against the oracle's restatement of the same lines, the initial
    conditions of a block = the block of the whole field, and the subdomain header of a block's files."""
    assert [crd.dims_create(n) for n in (1, 2, 3, 4, 6, 7, 8, 9, 12, 16)] == [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (7, 1), (4, 2), (3, 3), (4, 3), (4, 4)]
    rng = np.random.default_rng(11)
    for _ in range(60):
        nx, ny = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        d0, d1 = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        op = co.make_problem(co.FHN, co.TORUS, nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
        cover = np.zeros((ny, nx), dtype=int)
        for c0 in range(d0):
            for c1 in range(d1):
                is_, ie, js, je = crd.block_extents(nx, ny, c0, d0, c1, d1)
                sub = co.subproblem(op, d0, d1, c0, c1)
                assert (is_, ie, js, je) == (sub.is_, sub.ie, sub.js, sub.je)
                cover[js:je + 1, is_:ie + 1] += 1
        assert np.all(cover == 1)  # the blocks tile the grid
    with pytest.raises(crd._capi.CrdError):
        crd.block_extents(10, 10, 2, 2, 0, 1)
    # initial conditions of a block, all four programs' rules
    for model, surface, kw in (("fhn", "torus", {}), ("fhn", "flat", {}), ("goldbeter", "torus", dict(wave_inside=1)), ("goldbeter", "flat", {})):
        p = crd.make_params(model, surface, 37, 80.0, 20.0, 0.12, 1.25 if model == "fhn" else 0.4, ny=50)
        cfg = crd.run_config(p, wave_length=0.2, wave_width=0.5, **kw)
        whole = crd.initial_conditions(cfg)
        for c0, c1 in ((0, 0), (1, 2), (2, 1)):
            is_, ie, js, je = crd.block_extents(37, 50, c0, 3, c1, 3)
            assert np.array_equal(crd.initial_conditions(cfg, js, je, is_, ie), whole[js:je + 1, is_:ie + 1])
    # files of block (1, 0) of 2 x 2 = MPI rank 2: header "nx ny is ie js je xmin xmax tfinal", rows of nyl * nxl values
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    with crd.Writer(cfg, tmp_path, block=(1, 2, 0, 2)) as w:
        blk = crd.initial_conditions(cfg, 0, 19, 8, 15)
        w.write_row(blk)
    hdr = open(tmp_path / "FHNmodel_torus_subdomain.002.txt").read().split()
    assert [int(v) for v in hdr[:6]] == [16, 40, 8, 15, 0, 19]
    row = np.loadtxt(tmp_path / "FHNmodel_torus_u.002.txt", ndmin=2)
    assert row.shape == (1, 20 * 8) and np.array_equal(row[0].reshape(20, 8), blk[..., 0])


def test_seeded_sweep_of_host_side_rules_against_the_oracle():
    """300 seeded random parameter sets through the host-only entry points, each against the oracle's restatement: geometry
    scalars (incl. the `(long)(nx * R / r)` truncation and the flat `nx * (long)(L / W)` rule), slab extents that tile the
    grid, the halo plan's pairing, and the initial-condition rules of the four programs on random row ranges."""
    rng = np.random.default_rng(77)
    for case in range(300):
        model = ("fhn", "goldbeter")[int(rng.integers(2))]
        surface = ("torus", "flat")[int(rng.integers(2))]
        nx = int(rng.integers(4, 90))
        W = float(rng.choice([10.0, 20.0, 12.5, 7.0]))
        L = W * float(rng.choice([1.0, 2.0, 4.0, 5.0, 3.5, 2.2]))
        if surface == "torus" and L <= W * 1.01:
            L = 2.0 * W  # R = r is a singular (horn) torus
        beta = float(rng.uniform(0.8, 1.6)) if model == "fhn" else float(rng.uniform(0.2, 0.9))
        vary = int(rng.integers(2))
        ny_override = 0 if rng.integers(3) else int(rng.integers(8, 70))
        p = crd.make_params(model, surface, nx, L, W, 0.12, beta, ny=ny_override, vary_beta=vary, beta_min=0.7, beta_max=1.7)
        op = co.make_problem(co.FHN if model == "fhn" else co.GOLDBETER, co.TORUS if surface == "torus" else co.FLAT, nx, L, W, 0.12, beta, ny=ny_override,
                             vary_beta=vary, beta_min=0.7, beta_max=1.7)
        g = crd.grid_of(p)
        assert (g.nx, g.ny, g.dx, g.dy) == (op.nx, op.ny, op.dx, op.dy), (case, model, surface, nx, L, W)
        if g.ny < 8:
            continue
        # slabs tile the grid; the plan pairs every send with the matching receive of the neighbour
        n = int(rng.integers(1, min(6, g.ny // 2) + 1))
        ext = [crd.slab_extents(g.ny, k, n) for k in range(n)]
        assert ext[0][0] == 0 and ext[-1][1] == g.ny - 1 and all(ext[k][1] + 1 == ext[k + 1][0] for k in range(n - 1))
        depth = 1
        plans = [crd.halo_plan(k, n, ext[k][1] - ext[k][0] + 1, depth) for k in range(n)]
        for k in range(n):
            for is_send, peer, row_begin, _ in plans[k]:
                if is_send:  # every send has a receive from this slab waiting in the peer's plan
                    assert [q for q in plans[peer] if not q[0] and q[1] == k], (case, k, peer)
        # initial conditions on a random row range
        ic_type = int(rng.integers(3))
        cfg = crd.run_config(p, wave_length=float(rng.uniform(0.05, 0.4)), wave_width=float(rng.uniform(0.1, 0.9)), wave_inside=int(rng.integers(2)),
                             ic_type=ic_type)
        j0 = int(rng.integers(0, g.ny))
        j1 = int(rng.integers(j0, g.ny))
        got = crd.initial_conditions(cfg, j0, j1)  # every combination here is one a reference program runs: none may be refused
        want = co.initial_conditions(op, cfg.wave_length, cfg.wave_width, cfg.wave_inside, ic_type, steady_state=crd.steady_state(model, beta))
        if model == "goldbeter" and vary == 1 and ic_type == 2:
            # the random rule draws from rand() row by row from the FIRST row of the call (every reference rank seeds alike)
            assert got.shape == (j1 - j0 + 1, g.nx, 2) and np.all((got >= 0) & (got <= 1.4 + 1e-12))
            if j0 == 0:
                assert np.array_equal(got, want[: j1 + 1]), case
        else:
            assert np.array_equal(got, want[j0:j1 + 1]), (case, model, surface, vary, ic_type)


def test_initial_wave_and_beta_ramp_are_what_the_parameter_files_document():
    """The shipped parameter files describe their keys in comments (data/FHNmodelArgs.ini:24-41, data/GoldbeterModelArgs.ini:23-41),
    independently of the C++ source: waveLength / waveWidth are the initial wave segment's length / width "as a percentage of total
    length of torus (phi)" / "of total width of torus (theta)", waveInside centres it on the inside (theta = pi) or the outside
    (theta = 0) of the torus, and varyBeta varies beta "linearly over the surface" between betaMin and betaMax.  The host-side
    initial-condition rule and the row parameter the right-hand side applies are exactly that."""
    nx, ny = 400, 1600
    for inside in (0, 1):
        for wl, ww in ((0.1, 0.5), (0.25, 0.2)):
            p = crd.make_params("fhn", "torus", nx, 80.0, 20.0, 0.12, 1.25, ny=ny)
            y = crd.initial_conditions(crd.run_config(p, wave_length=wl, wave_width=ww, wave_inside=inside))
            us, vs = crd.steady_state("fhn", 1.25)
            wave = (y[..., 0] != us) | (y[..., 1] != vs)
            cols, rows = np.flatnonzero(wave.any(axis=0)), np.flatnonzero(wave.any(axis=1))
            assert wave.sum() == len(cols) * len(rows)  # a rectangle in (theta, phi)
            assert abs(len(rows) / ny - wl) <= 2.0 / ny and abs(len(cols) / nx - ww) <= 2.0 / nx  # the documented fractions of the surface
            theta = np.arange(nx) * 2 * np.pi / (nx - 1)
            if inside:  # centred on theta = pi
                assert abs(theta[cols].mean() - np.pi) <= 2 * np.pi / nx and np.all(np.diff(cols) == 1)
            else:  # centred on theta = 0 = 2 pi: the segment wraps around the seam
                assert cols[0] == 0 and cols[-1] == nx - 1 and abs(np.cos(theta[cols]).mean() - np.sinc(ww)) <= 0.02
    # beta varied linearly over phi between betaMin and betaMax: read off the FHN inhibitor equation, v' = 0.36 (u + b(phi))
    from oracle import crd_oracle as co

    op = co.make_problem(co.FHN, co.TORUS, 40, 80.0, 20.0, 0.12, 1.25, ny=160, vary_beta=1, beta_min=0.7, beta_max=1.7)
    yy = np.zeros((160, 40, 2))
    yy[..., 0], yy[..., 1] = 0.3, -0.2
    b = co.rhs(op, 1e9, yy)[:, 5, 1] / 0.36 - 0.3
    assert abs(b[0] - 0.7) <= 1e-12 and abs(b[-1] - 1.7) <= 1e-12 and np.max(np.abs(np.diff(b, 2))) <= 1e-12  # end points and linearity


def test_rccl_library_override_is_an_abi_call(tmp_path):
    """crd_comm_set_rccl_library names the file the nccl* entry points are bound from at first use (another RCCL build; the multi-process
    ring tests' stand-in).  No GPU needed to see the binding rule: a file that does not exist is reported by the first call that needs
    RCCL, with the file's name, as CRD_ERCCL; NULL / "" selects librccl.so.1 again; the Python package applies CRD_RCCL_LIBRARY."""
    import subprocess
    import sys

    code = (
        "import ctypes as C, crdmodel_amd as crd\n"
        "L = crd._capi.lib()\n"
        "buf = C.create_string_buffer(128)\n"
        "assert L.crd_comm_set_rccl_library(b'/nonexistent/librccl_other.so') == 0\n"
        "rc = L.crd_comm_unique_id(buf)\n"
        "print(rc, L.crd_last_error(None).decode())\n"
        "assert L.crd_comm_set_rccl_library(None) == 0 and L.crd_comm_set_rccl_library(b'') == 0\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    rc, message = r.stdout.strip().split(" ", 1)
    assert int(rc) == crd._capi.ERCCL and "/nonexistent/librccl_other.so" in message
    # ... and through the environment of a Python process
    r = subprocess.run([sys.executable, "-c", "import crdmodel_amd as crd\ntry:\n    crd.rccl_unique_id()\nexcept Exception as e:\n    print(e)\n"], capture_output=True, text=True,
                       cwd=ROOT, timeout=120, env=dict(os.environ, CRD_RCCL_LIBRARY="/nonexistent/from_env.so"))
    assert r.returncode == 0 and "/nonexistent/from_env.so" in r.stdout, (r.stdout, r.stderr)


def test_multi_step_kernels_issue_their_stores_whatever_the_execution_mask():
    """The vmcnt contract of the multi-step pipelines, checked on the device assembly of THIS build (round 6; the round-5 verdict's first
    item).  A ring slot's read waits with `s_waitcnt vmcnt(N)`, N = the vector-memory operations a wavefront issues between the slot's
    LDS-DMA fill and the read -- the row stores of four iterations among them (crd_fused_impl.h: kWaitSteady).  If the compiler wraps the
    stores in a skip branch on the execution mask (s_cbranch_execz), a wavefront without a storing lane has fewer operations in flight,
    the wait waits for nothing and the read can overtake the fill.  tools/kernel_regs.py counts, per kernel, the vector-memory regions
    an exec branch can skip without draining them; the build stops if a multi-step kernel has one, the table is compiled into the
    library (crd_launch_geometry::exec_skipped_vmem) and kept beside it (csrc/build/kernel_table.json), which is read here.  The
    same parser is then shown the shape it must catch."""
    import sys

    table = os.path.join(ROOT, "crdmodel_amd", "csrc", "build", "kernel_table.json")
    if not os.path.exists(table):
        pytest.skip("no kernel table beside the library (a build with KERNEL_TABLE=0)")
    doc = json.load(open(table))
    multi = [k for k in doc["kernels"] if k["steps"] >= 2]
    assert len(multi) >= 16 and all(k["exec_skipped_vmem"] == 0 for k in multi), [k for k in multi if k["exec_skipped_vmem"]]
    assert re.fullmatch(r"[0-9a-f]{16}", doc["digest"]) and crd._capi.lib().crd_kernel_table_digest().decode() == doc["digest"]
    # the headline kernel and the block-strip one are among them, at the register lines DESIGN.md states (three wavefronts per SIMD)
    by = {(k["precision"], k["model"], k["absorb"], k["cols"], k["nt"], k["steps"]): k for k in doc["kernels"] if k["embed"] == 0}
    assert by[("f64", 0, 0, 1, 1, 2)]["vgprs"] <= 168 and by[("f64", 0, 0, 1, 1, 2)]["wavefronts_per_simd"] == 3
    assert by[("f64", 1, 0, 1, 1, 2)]["vgprs"] <= 168 and by[("f64", 1, 0, 1, 1, 2)]["wavefronts_per_simd"] == 3
    # The micro-tricks the hot loops lean on, watched where they would show if a compiler stopped honouring them (the round-5 verdict's
    # "compiler-fragile micro-tricks"): plain register moves in the steady-state loop.  The three-address stage updates (stage_fma) and
    # the kinetics' constant held in vector registers (in_vector_registers) keep the fp64 multi-step loops at 4 moves per trip (the
    # compiler's own forms: 10 + 8 per iteration); the packed fp32 loops' per-column lane shifts (rhs_lane, built without the SLP
    # vectoriser) keep theirs at 4 (assembled shifted pairs: 68 - 100).  And no scratch, at the occupancy DESIGN.md states.
    for key, vgpr_max, waves in ((("f64", 0, 0, 1, 1, 2), 168, 3), (("f64", 0, 0, 1, 1, 3), 256, 2), (("f64", 1, 0, 1, 1, 2), 168, 3), (("f32", 0, 0, 2, 1, 2), 168, 3),
                                 (("f32", 0, 0, 2, 1, 3), 256, 2)):
        k = by[key]
        assert k["loop"]["moves"] <= 8 and k["scratch_bytes"] == 0 and k["vgprs"] <= vgpr_max and k["wavefronts_per_simd"] == waves, (key, k)
    assert by[("f64", 0, 0, 1, 1, 3)]["loop"]["valu"] <= 870 and by[("f32", 0, 0, 2, 1, 3)]["loop"]["valu"] <= 770 and by[("f64", 0, 0, 1, 1, 2)]["loop"]["valu"] <= 545
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_regs
    finally:
        sys.path.pop(0)
    skipped_store = """
	s_and_saveexec_b64 s[6:7], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	buffer_store_dwordx2 v[0:1], v2, s[8:11], 0 offen nt
	buffer_store_dwordx2 v[4:5], v2, s[12:15], 0 offen nt
.LBB0_2:
	s_or_b64 exec, exec, s[6:7]
	s_waitcnt vmcnt(20)
	ds_read_b64 v[0:1], v3
""".splitlines()
    assert len(kernel_regs.exec_skipped_vmem(skipped_store)) == 1
    drained = [ln for ln in skipped_store]
    drained.insert(6, "\ts_waitcnt vmcnt(0)")  # a region that drains what it issued: skipped or not, nothing of it is in flight behind it
    assert kernel_regs.exec_skipped_vmem(drained) == []
    unconditional = [ln for ln in skipped_store if "exec" not in ln]
    assert kernel_regs.exec_skipped_vmem(unconditional) == []
    # The other contract with the compiler (round 6): the block-strip kernels issue the LDS reads of the neighbours' edge values in one asm
    # block at the end of an iteration and wait for them in another at the start of the next; in between nothing may touch the registers
    # the reads are in flight into -- the compiler believes them written and is free to copy them (an experiment that did the same for the
    # ring reads got copies at the loop's back edge and results that changed from run to run).  Checked on the assembly of every build
    # (kernel_regs.py: async_lds_read_hazards; the build stops on one), and here on the shape it must catch.
    assert all(k.get("async_lds_read_hazards", 0) == 0 for k in doc["kernels"])
    in_flight = """
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(44)
	ds_read_b64 v[10:11], v200 offset:0
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_fma_f64 v[20:21], v[10:11], v[30:31], v[40:41]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	s_barrier
	ds_read_b128 v[46:49], v252 offset:0
	;;#ASMEND
	s_add_i32 s20, s20, 1
	s_cmp_lt_i32 s20, s21
	s_cbranch_scc1 .LBB0_1
	s_endpgm
""".splitlines()
    assert kernel_regs.async_lds_read_hazards(in_flight) == []  # (nothing between the issue and the waiting block at the loop's head)
    copied = list(in_flight)
    copied.insert(14, "\tv_mov_b64_e32 v[60:61], v[48:49]")  # a copy of a register with a read in flight, in front of the back edge
    hazards = kernel_regs.async_lds_read_hazards(copied)
    assert len(hazards) == 1 and hazards[0][1] in (48, 49)
