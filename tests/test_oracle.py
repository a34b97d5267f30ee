"""The oracle checks itself: C restatement vs golden vectors, vs the independent numpy restatement, vs analytic
known answers the reference's code implies, and decomposition invariance (what MPI delivers at np in {1,2,4}).

PARITY UNPINNED against reference-run output (the reference cannot be built here; it ships no tests or fixtures)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden, oracle_problem, rel_err
from oracle import crd_oracle as co
from oracle import crd_oracle_np as cn


def np_kwargs(meta):
    return dict(beta=meta["beta"], vary_beta=meta["vary_beta"], beta_min=meta["beta_min"], beta_max=meta["beta_max"],
                t_boundary=meta["t_boundary"], just_diffusion=meta["just_diffusion"])


@pytest.mark.parametrize("name", golden_names("rhs_"))
def test_c_oracle_reproduces_golden_rhs(name):
    meta, arr = load_golden(name)
    p = oracle_problem(meta)
    assert (p.nx, p.ny) == (meta["nx"], meta["ny"])
    assert np.array_equal(co.rhs(p, meta["t_absorbing"], arr["y"]), arr["ydot_absorbing"])
    assert np.array_equal(co.rhs(p, meta["t_free"], arr["y"]), arr["ydot_free"])
    assert np.array_equal(co.rhs(p, meta["t_free"], arr["y"], nthreads=3), arr["ydot_free"])  # OpenMP split changes nothing


@pytest.mark.parametrize("name", golden_names("rhs_"))
def test_numpy_restatement_agrees(name):
    """Two independently written restatements (C loops with halo strips; numpy with np.roll) agree to round-off."""
    meta, arr = load_golden(name)
    g = cn.geometry(meta["surface"], meta["surface_length"], meta["surface_width"], meta["nx"], meta["ny_override"])
    assert g["ny"] == meta["ny"]
    for key, t in (("ydot_absorbing", meta["t_absorbing"]), ("ydot_free", meta["t_free"])):
        du, dv = cn.rhs(meta["model"], meta["surface"], g, meta["diffusion"], t, arr["y"][..., 0], arr["y"][..., 1], **np_kwargs(meta))
        assert rel_err(du, arr[key][..., 0]) <= 2e-15
        assert rel_err(dv, arr[key][..., 1]) <= 2e-15


@pytest.mark.parametrize("name", golden_names("rk4_"))
def test_rk4_golden_and_numpy(name):
    meta, arr = load_golden(name)
    p = oracle_problem(meta)
    n = meta["snapshots"][0]
    y = co.rk4(p, arr["y0"], 0.0, meta["dt"], n)
    assert np.array_equal(y, arr["y_%d" % n])
    g = cn.geometry(meta["surface"], meta["surface_length"], meta["surface_width"], meta["nx"], meta["ny_override"])
    short = min(n, 20)
    u, v = cn.rk4(meta["model"], meta["surface"], g, meta["diffusion"], arr["y0"][..., 0], arr["y0"][..., 1], 0.0, meta["dt"], short, **np_kwargs(meta))
    yc = co.rk4(p, arr["y0"], 0.0, meta["dt"], short)
    assert rel_err(u, yc[..., 0]) <= 1e-13 and rel_err(v, yc[..., 1]) <= 1e-13


@pytest.mark.parametrize("dims", [(1, 1), (2, 1), (1, 2), (2, 2), (1, 4), (1, 8), (3, 2)])
@pytest.mark.parametrize("name", ["rhs_fhn_torus_ragged", "rhs_goldbeter_flat"])
def test_decomposition_invariance(name, dims):
    """f through a d0 x d1 block decomposition with periodic halos == f on one block, bit for bit; (2,2) is the reference's
    own `mpirun -np 4` layout, (1,G) the phi-slab layout of the GPU build."""
    meta, arr = load_golden(name)
    p = oracle_problem(meta)
    assert np.array_equal(co.rhs(p, meta["t_absorbing"], arr["y"], *dims), arr["ydot_absorbing"])


def test_exchange_strips_are_the_periodic_neighbours():
    """Exchange() on one rank: the W strip is the subdomain's own last column, the S strip its last row (:854-900)."""
    meta, arr = load_golden("rhs_fhn_torus_ragged")
    p = oracle_problem(meta)
    w, e, s, n = co.pack_edges(p, arr["y"])
    assert np.array_equal(w.reshape(-1, 2), arr["y"][:, -1, :]) and np.array_equal(e.reshape(-1, 2), arr["y"][:, 0, :])
    assert np.array_equal(s.reshape(-1, 2), arr["y"][-1, :, :]) and np.array_equal(n.reshape(-1, 2), arr["y"][0, :, :])
    assert np.array_equal(co.rhs_subdomain(p, 50.0, arr["y"], w, e, s, n), arr["ydot_free"])


# ---- analytic known answers ---------------------------------------------------------------------------------

@pytest.mark.parametrize("surface,L,W", [("torus", 80.0, 20.0), ("torus", 40.0, 20.0), ("flat", 80.0, 20.0)])
@pytest.mark.parametrize("beta", [0.7, 1.25])
def test_fhn_stable_state_is_a_fixed_point(surface, L, W, beta):
    """Us = -beta, Vs = beta^3 - 3 beta (src/FHNmodel_torus.cpp:242-244): a uniform field at the stable state has f = 0."""
    p = co.make_problem(co.FHN, {"torus": co.TORUS, "flat": co.FLAT}[surface], 24, L, W, 0.12, beta)
    us, vs = co.fhn_steady(beta)
    y = np.empty((p.ny, p.nx, 2))
    y[..., 0], y[..., 1] = us, vs
    assert np.max(np.abs(co.rhs(p, 0.0, y))) <= 1e-13


@pytest.mark.parametrize("beta", [0.14, 0.4, 0.6, 1.0])
def test_goldbeter_fixed_point(beta):
    zs, ys = co.goldbeter_steady(beta)
    assert zs == pytest.approx((1.0 + 7.3 * beta) / 10.0, rel=1e-15)
    zn, yn = cn.goldbeter_steady(beta)
    assert (zs, ys) == pytest.approx((zn, yn), rel=1e-13)
    p = co.make_problem(co.GOLDBETER, co.TORUS, 16, 40.0, 20.0, 0.12, beta)
    y = np.empty((p.ny, p.nx, 2))
    y[..., 0], y[..., 1] = zs, ys
    assert np.max(np.abs(co.rhs(p, 0.0, y))) <= 2e-12


@pytest.mark.parametrize("m", [1, 3])
def test_flat_laplacian_eigenfunction(m):
    """cos(2 pi m i / nx) is an exact eigenvector of the periodic second difference: udot = cu1 (2 cos(2 pi m/nx) - 2) u
    (src/FHNmodel_flat.cpp:489-500, index -1 wrapping to nx-1)."""
    nx = 32
    p = co.make_problem(co.GOLDBETER, co.FLAT, nx, 20.0, 20.0, 0.12, 0.4, just_diffusion=1)
    i = np.arange(nx)
    u = np.broadcast_to(np.cos(2 * np.pi * m * i / nx), (p.ny, nx))
    y = np.stack([u, np.ones_like(u)], axis=-1).copy()
    lam = 0.12 / p.dx / p.dx * (2 * np.cos(2 * np.pi * m / nx) - 2)
    ydot = co.rhs(p, 0.0, y)
    assert rel_err(ydot[..., 0], lam * u) <= 1e-13
    assert np.all(ydot[..., 1] == 0.0)


def test_torus_phi_eigenfunction():
    """For u = cos(2 pi m j / ny) (theta-independent) only the phi term survives:
    udot = D / (R + r cos theta_i)^2 / dy^2 * (2 cos(2 pi m / ny) - 2) u  (src/FHNmodel_torus.cpp:537)."""
    p = co.make_problem(co.GOLDBETER, co.TORUS, 20, 80.0, 20.0, 0.12, 0.4, just_diffusion=1)
    m = 2
    j = np.arange(p.ny)
    u = np.broadcast_to(np.cos(2 * np.pi * m * j / p.ny)[:, None], (p.ny, p.nx))
    y = np.stack([u, np.zeros_like(u)], axis=-1).copy()
    theta = np.arange(p.nx) * p.dx
    expect = 0.12 / (p.R + p.r * np.cos(theta)) ** 2 / p.dy ** 2 * (2 * np.cos(2 * np.pi * m / p.ny) - 2) * u
    assert rel_err(co.rhs(p, 0.0, y)[..., 0], expect) <= 1e-13


def test_torus_theta_operator_is_the_curvilinear_laplacian():
    """The theta part is (1/r^2) u_tt - sin t / (r (R + r cos t)) u_t: second-order accurate on a smooth periodic field."""
    errs = []
    for nx in (64, 128):
        p = co.make_problem(co.GOLDBETER, co.TORUS, nx, 80.0, 20.0, 1.0, 0.4, ny=8, just_diffusion=1)
        # sample on the periodic points the stencil actually couples: theta_i = i dx with dx = 2 pi/(nx-1) wraps with a
        # duplicated seam, so use a field that is smooth in INDEX space: k = 2 pi i / nx
        i = np.arange(nx)
        k = 2 * np.pi * i / nx
        u = np.broadcast_to(np.sin(k), (p.ny, nx))
        y = np.stack([u, np.zeros_like(u)], axis=-1).copy()
        got = co.rhs(p, 0.0, y)[0, :, 0]
        theta = i * p.dx
        s = 2 * np.pi / nx / p.dx  # d k / d theta
        exact = (1 / p.r ** 2) * (-s * s * np.sin(k)) + (-np.sin(theta) / (p.r * (p.R + p.r * np.cos(theta)))) * (s * np.cos(k))
        errs.append(np.max(np.abs(got - exact)))
    assert errs[1] < errs[0] / 3.5  # ~4x per mesh doubling


def kinetics_jacobian_trace(f_uniform, state, eps=1e-6):
    """Trace of the 2 x 2 Jacobian of the reaction terms at `state`, by central differences of f on uniform fields (where the
    diffusion term vanishes identically)."""
    s0, s1 = state
    d0 = (f_uniform(s0 + eps, s1)[0] - f_uniform(s0 - eps, s1)[0]) / (2 * eps)
    d1 = (f_uniform(s0, s1 + eps)[1] - f_uniform(s0, s1 - eps)[1]) / (2 * eps)
    return d0 + d1


def hopf_points(model, f_uniform_for_beta, brackets):
    """beta at which the steady state of the kinetics changes stability (trace of the Jacobian = 0)."""
    from scipy.optimize import brentq

    return [brentq(lambda b: kinetics_jacobian_trace(f_uniform_for_beta(b), co.steady(model, b)), lo, hi, xtol=1e-9) for lo, hi in brackets]


def oracle_uniform_f(model, beta):
    p = co.make_problem(model, co.FLAT, 8, 20.0, 20.0, 0.12, beta)

    def f(s0, s1):
        y = np.empty((p.ny, p.nx, 2))
        y[..., 0], y[..., 1] = s0, s1
        return co.rhs(p, 0.0, y)[3, 3]

    return f


def test_known_answers_the_reference_states_about_its_kinetics():
    """The only numbers the reference tree holds about this path's results: the Goldbeter parameter sets say the model is
    "oscillatory when 0.28895 < beta < 0.77427" (data/GoldbeterModelArgs.ini:25, data/temp.ini:21; the torus-mapping script
    marks 0.289 and 0.774, util/GoldbeterModel/MapOutputToTorus.py:60-63), and the FHN utilities put the Hopf bifurcation
    at beta = 1 (util/FHNmodel/plot_FHNmodel_torus.py:90-92).  The restated kinetics reproduce both to the digits given."""
    lo, hi = hopf_points(co.GOLDBETER, lambda b: oracle_uniform_f(co.GOLDBETER, b), [(0.2, 0.4), (0.6, 0.9)])
    assert abs(lo - 0.28895) <= 5e-6 and abs(hi - 0.77427) <= 5e-6, (lo, hi)
    (one,) = hopf_points(co.FHN, lambda b: oracle_uniform_f(co.FHN, b), [(0.5, 1.5)])
    assert abs(one - 1.0) <= 1e-7, one
    # ... and on the right sides: "oscillatory for beta < 1, stable for beta > 1" (data/FHNmodelArgs.ini:24), oscillatory inside
    # the Goldbeter window only
    tr = lambda model, b: kinetics_jacobian_trace(oracle_uniform_f(model, b), co.steady(model, b))
    assert tr(co.FHN, 0.9) > 0 > tr(co.FHN, 1.1)
    assert tr(co.GOLDBETER, 0.5) > 0 and tr(co.GOLDBETER, 0.2) < 0 and tr(co.GOLDBETER, 0.9) < 0


def test_absorbing_rows_rule():
    """Rows j = 0 and j = ny-1 get f = 0 for both variables only while t < tBoundary (strict), :643-653."""
    meta, arr = load_golden("rhs_fhn_torus")
    p = oracle_problem(meta)
    tb = meta["t_boundary"]
    before, at = co.rhs(p, np.nextafter(tb, 0.0), arr["y"]), co.rhs(p, tb, arr["y"])
    assert np.all(before[0] == 0) and np.all(before[-1] == 0) and np.any(before[1] != 0)
    assert np.array_equal(at, arr["ydot_free"]) and np.any(at[0] != 0)
    assert np.array_equal(before[1:-1], at[1:-1])  # the neighbours still read the boundary rows normally


def test_goldbeter_just_diffusion_skips_absorbing_rows():
    meta, arr = load_golden("rhs_goldbeter_torus_justdiffusion")
    assert np.array_equal(arr["ydot_absorbing"], arr["ydot_free"])
    assert np.all(arr["ydot_free"][..., 1] == 0.0)


def test_ny_truncation_table():
    """ny = (long)(NX * (R / r)) truncates (src/FHNmodel_torus.cpp:193): L=100, W=20, nx=100 gives 499, not 500."""
    import json, os
    from conftest import GOLDEN

    geo = json.load(open(os.path.join(GOLDEN, "geometry.json")))
    by = {(e["surface"], e["L"], e["W"], e["nx"]): e["ny"] for e in geo["ny"]}
    assert by[("torus", 100.0, 20.0, 100)] == 499 and by[("torus", 100.0, 20.0, 400)] == 1999
    assert by[("torus", 80.0, 20.0, 400)] == 1600 and by[("flat", 90.0, 20.0, 100)] == 400
    for e in geo["ny"]:
        g = cn.geometry(e["surface"], e["L"], e["W"], e["nx"])
        assert (g["ny"], g["dx"], g["dy"]) == (e["ny"], e["dx"], e["dy"])


def test_adaptive_restatement_is_sane():
    """The CPU restatement of the error-controlled stepper (used to check libcrd's crd_integrate_adaptive): lands on tout,
    rejects when started far above the stable step, and converges to a fine fixed-step RK4 solution at the tolerance's scale."""
    meta, arr = load_golden("rk4_fhn_torus_outside")
    p = oracle_problem(meta)
    tout = 1.0
    y, st = co.integrate_adaptive(p, arr["y0"], 0.0, tout, h0=2.0)
    assert st["t"] == tout and st["rejected"] >= 1 and st["accepted"] == len(st["steps"]) and abs(sum(st["steps"]) - tout) < 1e-12
    fine = co.rk4(p, arr["y0"], 0.0, tout / 2000, 2000)
    assert rel_err(y, fine) <= 1e-3
    y2, st2 = co.integrate_adaptive(p, arr["y0"], 0.0, tout, h0=2.0, rtol=1e-8, atol=1e-12)
    assert st2["accepted"] > st["accepted"] and rel_err(y2, fine) < 0.05 * rel_err(y, fine) + 1e-9


def test_arkode_table_satisfies_its_order_conditions():
    """oracle/arkode_erk.py restates ARKode's default explicit fourth-order table (ARK_ZONNEVELD_5_3_4) from its documentation,
    without the library to check against; what can be checked is that the numbers ARE a 5-stage pair of orders 4 and 3: row
    sums equal the abscissae, all eight order conditions up to order 4 for b, all four up to order 3 for the embedding -- and that
    the embedding is NOT of order 4 (else the difference would estimate nothing)."""
    from oracle import arkode_erk as ark

    t = ark.ZONNEVELD_5_3_4
    A, b, b2, c = np.array(t["A"]), np.array(t["b"]), np.array(t["b2"]), np.array(t["c"])
    assert np.allclose(A.sum(axis=1), c, atol=1e-15) and np.all(np.triu(A) == 0.0) and (t["q"], t["p"]) == (4, 3)
    Ac = A @ c
    conditions = lambda w: [w.sum() - 1, w @ c - 1 / 2, w @ c ** 2 - 1 / 3, w @ Ac - 1 / 6,  # noqa: E731
                            w @ c ** 3 - 1 / 4, w @ (c * Ac) - 1 / 8, w @ (A @ c ** 2) - 1 / 12, w @ (A @ Ac) - 1 / 24]
    assert np.allclose(conditions(b), 0.0, atol=1e-15)
    assert np.allclose(conditions(b2)[:4], 0.0, atol=2e-15) and max(abs(v) for v in conditions(b2)[4:]) > 1e-2
    assert np.array_equal(b[:4], np.array([1, 2, 2, 1]) / 6.0) and b[4] == 0.0  # the propagated solution is classical RK4
    # the error weights the kernel uses: b - b2 = (2/3, -2, -2, -2, 16/3)
    assert np.allclose(b - b2, [2 / 3, -2, -2, -2, 16 / 3], atol=1e-15)


def test_arkode_restatement_behaves_like_the_documented_controller():
    """ArkodeErk around the oracle's f(): the propagated solution of one step equals classical RK4's, the error estimate scales
    like h^4, ARK_NORMAL output (no step is shortened; outputs inside a finished step re-interpolate without stepping; ARKode's
    Hermite form equals the textbook one), the dead band keeps h constant for a suggested growth within [1, 1.5], a failed step
    is followed by one of the same size, and the result converges to a fine fixed-step solution at the tolerance's scale."""
    from oracle import arkode_erk as ark

    meta, arr = load_golden("rk4_fhn_torus_outside")
    p = oracle_problem(meta)
    y0 = arr["y0"]
    f = lambda t, y: co.rhs(p, t, y)  # noqa: E731
    h = 0.02
    ynew, err = ark.erk_attempt(f, 0.0, y0, h)
    assert rel_err(ynew, co.rk4(p, y0, 0.0, h, 1)) <= 1e-14
    _, err2 = ark.erk_attempt(f, 0.0, y0, h / 2)
    assert 8.0 < np.abs(err).max() / np.abs(err2).max() < 32.0  # local error of the third-order embedding: h^4
    tau = -0.37
    a = [np.random.default_rng(k).standard_normal(5) for k in range(4)]
    assert np.allclose(ark.hermite_arkode(tau, h, *a), co.hermite(1.0 + tau, h, a[0], a[1], a[2], a[3]), rtol=1e-13, atol=1e-15)
    # controller rules
    assert ark.pid_eta(0.1, [0.5, 1.0, 1.0], 20.0) == 1.0                      # suggested growth 1.07: inside the dead band
    assert ark.pid_eta(0.1, [1e-6, 1.0, 1.0], 20.0) == pytest.approx(0.96 * (1e-6) ** (-0.58 / 3))  # 13.9x
    assert ark.pid_eta(0.1, [1e-12, 1.0, 1.0], 20.0) == 20.0 and ark.pid_eta(0.1, [1e-12, 1.0, 1.0], 1e4) == pytest.approx(0.96 * (1e-10) ** (-0.58 / 3))
    assert ark.pid_eta(0.1, [1e30, 1.0, 1.0], 1.0) == pytest.approx(0.1, rel=1e-14) and ark.pid_eta(0.1, [3.0, 1.0, 1.0], 1.0) < 1.0
    assert ark.pid_eta(0.1, [1e-6, 1.0, 1.0], 20.0, h_max=0.12) == pytest.approx(1.2)
    integ = ark.ArkodeErk(p, 0.0, y0)
    y1, st1 = integ.evolve(1.0)
    assert st1["t_internal"] >= 1.0 and abs(sum(integ.steps) - st1["t_internal"]) < 1e-12 and integ.steps[0] < 0.01 < max(integ.steps)
    y1b, st1b = integ.evolve(st1["t_internal"])  # exactly the end of the step taken: its state, no new step
    assert st1b["accepted"] == 0 and np.array_equal(y1b, integ.y)
    y3, st3 = integ.evolve(3.0)
    assert integ.netf >= 1  # this grid is diffusion-limited: error control alone finds the stability bound by failing
    runs = [len(list(g)) for _, g in __import__("itertools").groupby(integ.steps)]
    assert max(runs) >= 5  # the dead band: long runs of identical steps
    fine = co.rk4(p, y0, 0.0, 3.0 / 3000, 3000)
    assert rel_err(y3, fine) <= 2e-3
    tight = ark.ArkodeErk(p, 0.0, y0, rtol=1e-8, atol=1e-12)
    y3t, _ = tight.evolve(3.0)
    assert tight.nst > integ.nst and rel_err(y3t, fine) < 0.05 * rel_err(y3, fine) + 1e-9


def curvature_reference_tables(theta, r, R, substeps=16):
    """What the reference's utilities state about the torus, turned into the two theta-dependent coefficients of the diffusion
    operator WITHOUT using the operator's own formulas.  util/PlotGaussianAndCoupling.py:11-12 (= util/GenCurvatureCoupling.py:87)
    give the Gaussian curvature G(theta) = cos(theta) / (r (R + r cos(theta))).  On a surface of revolution with metric
    ds^2 = r^2 dtheta^2 + rho(theta)^2 dphi^2 the Laplace-Beltrami operator is
        (1/r^2) u_thth + b(theta) u_th + rho(theta)^-2 u_phph,     b = rho' / (rho r^2),     G = -rho'' / (rho r^2),
    so G alone, with rho(0) = R + r (the outer equator: (surfaceLength + surfaceWidth) / 2 pi) and rho'(0) = 0 (symmetry), fixes
    both: rho'' = -G rho r^2 is integrated here (classical RK4, `substeps` per grid interval) up to every theta_i.
    Returns (b(theta_i), rho(theta_i)^-2)."""
    def G(t):
        return np.cos(t) / (r * (R + r * np.cos(t)))

    def rhs(t, y):
        return np.array([y[1], -G(t) * y[0] * r * r])

    y, t = np.array([R + r, 0.0]), 0.0
    b, inv_rho2 = np.empty_like(theta), np.empty_like(theta)
    for i, ti in enumerate(theta):
        if ti > t:
            h = (ti - t) / substeps
            for _ in range(substeps):
                k1 = rhs(t, y)
                k2 = rhs(t + h / 2, y + h / 2 * k1)
                k3 = rhs(t + h / 2, y + h / 2 * k2)
                k4 = rhs(t + h, y + h * k3)
                y = y + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
                t += h
        b[i] = y[1] / (y[0] * r * r)
        inv_rho2[i] = 1.0 / (y[0] * y[0])
    return b, inv_rho2


def coupling_strength_as_the_reference_plots_it(theta, r, R):
    """util/PlotGaussianAndCoupling.py:15-19 (= util/GenCurvatureCoupling.py:30-40,90): the "coupling strength" of Kneer et al. in
    the torus's isothermal coordinates, C = 10 (cosh(eta) - cos(theta_i))^2 / a^2 with a = sqrt(R^2 - r^2), eta = atanh(a / R),
    theta_i = arccos(R/r - a^2 / (r (R + r cos(theta)))).  It is the inverse of the conformal factor, hence proportional to the
    phi-phi coefficient of the Laplace-Beltrami operator, (R + r cos(theta))^-2 (the factor is 10 a^2 / r^2)."""
    a = np.sqrt(R * R - r * r)
    eta = np.arctanh(a / R)
    theta_i = np.arccos(np.clip(R / r - a * a / (r * (R + r * np.cos(theta))), -1.0, 1.0))
    return 10.0 * (np.cosh(eta) - np.cos(theta_i)) ** 2 / (a * a)


def coefficients_implied_by_f(f, g, diffusion):
    """The two theta-dependent coefficients as `f` applies them, recovered from two evaluations of the diffusion-only right-hand
    side treated as a black box: on u = sin(theta_i) (no phi dependence) the phi term vanishes and
        f_u / D = (1/r^2) (u_E - 2u + u_W)/dx^2 + b_i (u_E - u_W)/(2 dx)     =>  b_i;
    on u = sin(phi_j) the theta terms vanish and f_u / D = c_i (u_N - 2u + u_S)/dy^2  =>  c_i.  The differences are formed from
    the very field values handed to f, so nothing is lost to truncation; columns / rows next to the periodic seam are left out."""
    nx, ny = int(g.nx), int(g.ny)
    theta, phi = g.xmin + np.arange(nx) * g.dx, g.ymin + np.arange(ny) * g.dy
    y = np.zeros((ny, nx, 2))
    y[..., 0] = np.sin(theta)[None, :]
    y[..., 1] = 0.7
    fu = f(y)[ny // 2, :, 0] / diffusion
    u = np.sin(theta)
    i = np.arange(2, nx - 2)
    i = i[np.abs(np.cos(theta[i])) > 0.2]
    b = (fu[i] - (u[i + 1] - 2 * u[i] + u[i - 1]) / (g.dx * g.dx) / (g.r * g.r)) / ((u[i + 1] - u[i - 1]) / (2 * g.dx))
    y[..., 0] = np.sin(phi)[:, None]
    fphi = f(y)[:, :, 0] / diffusion
    w = np.sin(phi)
    j = np.arange(2, ny - 2)
    j = j[np.abs(w[j]) > 0.3][0]
    c = fphi[j, :] / ((w[j + 1] - 2 * w[j] + w[j - 1]) / (g.dy * g.dy))
    return i, theta, b, c


@pytest.mark.parametrize("L,W", [(80.0, 20.0), (40.0, 20.0)])
def test_diffusion_operator_reproduces_the_gaussian_curvature_the_reference_states(L, W):
    """A known answer for the DIFFUSION part of `f` from the reference tree itself: its plotting utilities state the torus's
    Gaussian curvature, G = cos(theta) / (r (R + r cos(theta))) (util/PlotGaussianAndCoupling.py:11-12, util/GenCurvatureCoupling.py:87,
    for the two tori it uses: surfaceLength 80 and 40, surfaceWidth 20).  G fixes the theta-advection coefficient and the phi-phi
    coefficient of the Laplace-Beltrami operator (curvature_reference_tables); the coefficients the restated `f` applies
    (src/FHNmodel_torus.cpp:527-615 = src/GoldbeterModel_torus.cpp:571-659) must be those -- in the C restatement and in the numpy one."""
    nx, ny, D = 400, 24, 0.12
    op = co.make_problem(co.GOLDBETER, co.TORUS, nx, L, W, D, 0.4, ny=ny, just_diffusion=1)
    gn = cn.geometry("torus", L, W, nx, ny)

    def f_numpy(y):
        du, dv = cn.rhs("goldbeter", "torus", gn, D, 0.0, y[..., 0], y[..., 1], beta=0.4, just_diffusion=1)
        return np.stack([du, dv], axis=-1)

    for f in (lambda y: co.rhs(op, 0.0, y), f_numpy):
        i, theta, b, c = coefficients_implied_by_f(f, op, D)
        b_ref, c_ref = curvature_reference_tables(theta, op.r, op.R)
        assert np.max(np.abs(b - b_ref[i])) <= 1e-9 * np.max(np.abs(b_ref)), float(np.max(np.abs(b - b_ref[i])))
        assert np.max(np.abs(c[2:-2] / c_ref[2:-2] - 1.0)) <= 1e-9
        # ... and the phi-phi coefficient has the shape of the "coupling strength" the same utilities plot
        cs = coupling_strength_as_the_reference_plots_it(theta, op.r, op.R)
        mid = len(theta) // 3
        assert np.max(np.abs((c[2:-2] / c[mid]) / (cs[2:-2] / cs[mid]) - 1.0)) <= 1e-11
