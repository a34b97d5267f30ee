"""ctypes front end of the C oracle (oracle/crd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(crdmodel_amd/, libcrd.so) never does.  PARITY UNPINNED against reference-run output -- see crd_oracle.h.

State vectors cross this interface in the reference's own layout: AoS [u, v] pairs, theta (i) fastest,
IDX(i, j) = 2 i + 2 j nxl (/root/reference/src/FHNmodel_torus.cpp:60), i.e. numpy shape (nyl, nxl, 2).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcrd_oracle.so")

FHN, GOLDBETER = 0, 1
TORUS, FLAT = 0, 1


class Problem(C.Structure):
    """Mirror of crd_oracle_problem."""

    _fields_ = [
        ("model", C.c_int), ("surface", C.c_int),
        ("nx", C.c_long), ("ny", C.c_long),
        ("is_", C.c_long), ("ie", C.c_long), ("js", C.c_long), ("je", C.c_long),
        ("dx", C.c_double), ("dy", C.c_double),
        ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
        ("R", C.c_double), ("r", C.c_double),
        ("diff", C.c_double),
        ("beta", C.c_double), ("beta_min", C.c_double), ("beta_max", C.c_double),
        ("vary_beta", C.c_int), ("just_diffusion", C.c_int),
        ("t_boundary", C.c_double),
    ]

    @property
    def nxl(self):
        return self.ie - self.is_ + 1

    @property
    def nyl(self):
        return self.je - self.js + 1


class IC(C.Structure):
    _fields_ = [("wave_length", C.c_double), ("wave_width", C.c_double), ("wave_inside", C.c_int),
                ("ic_type", C.c_int), ("s0", C.c_double), ("s1", C.c_double)]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only; seconds)."""
    src = [os.path.join(_HERE, f) for f in ("crd_oracle.c", "crd_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        lp = C.POINTER(C.c_long)
        L.crd_oracle_geometry.argtypes = [C.c_int, C.c_double, C.c_double, C.c_long, C.c_long, C.POINTER(Problem)]
        L.crd_oracle_decomp.argtypes = [C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, lp, lp, lp, lp]
        L.crd_oracle_decomp.restype = None
        L.crd_oracle_fhn_steady.argtypes = [C.c_double, dp, dp]
        L.crd_oracle_fhn_steady.restype = None
        L.crd_oracle_goldbeter_steady.argtypes = [C.c_double, dp, dp]
        L.crd_oracle_initial_conditions.argtypes = [C.POINTER(Problem), C.POINTER(IC), dp]
        L.crd_oracle_rhs_subdomain.argtypes = [C.POINTER(Problem), C.c_double, dp, dp, dp, dp, dp, dp, C.c_int]
        L.crd_oracle_pack_edges.argtypes = [C.POINTER(Problem), dp, dp, dp, dp, dp]
        L.crd_oracle_pack_edges.restype = None
        L.crd_oracle_rhs_global.argtypes = [C.POINTER(Problem), C.c_double, dp, dp, C.c_int, C.c_int, C.c_int]
        L.crd_oracle_rk4.argtypes = [C.POINTER(Problem), dp, C.c_double, C.c_double, C.c_long, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_problem(model, surface, nx, surface_length, surface_width, diffusion, beta, *, ny=0, beta_min=0.0,
                 beta_max=0.0, vary_beta=0, just_diffusion=0, t_boundary=0.0):
    """Whole-domain problem; ny=0 derives ny the reference's way, ny>0 is the phiMesh override."""
    p = Problem()
    rc = lib().crd_oracle_geometry(surface, surface_length, surface_width, nx, ny, C.byref(p))
    if rc != 0:
        raise ValueError("crd_oracle_geometry failed")
    p.model = model
    p.diff = diffusion
    p.beta, p.beta_min, p.beta_max = beta, beta_min, beta_max
    p.vary_beta, p.just_diffusion = vary_beta, just_diffusion
    p.t_boundary = t_boundary
    return p


def subproblem(p, d0, d1, c0, c1):
    """Copy of p restricted to block (c0, c1) of a d0 x d1 process grid."""
    q = Problem.from_buffer_copy(p)
    is_, ie, js, je = C.c_long(), C.c_long(), C.c_long(), C.c_long()
    lib().crd_oracle_decomp(p.nx, p.ny, d0, d1, c0, c1, C.byref(is_), C.byref(ie), C.byref(js), C.byref(je))
    q.is_, q.ie, q.js, q.je = is_.value, ie.value, js.value, je.value
    return q


def fhn_steady(beta):
    a, b = C.c_double(), C.c_double()
    lib().crd_oracle_fhn_steady(beta, C.byref(a), C.byref(b))
    return a.value, b.value


def goldbeter_steady(beta):
    a, b = C.c_double(), C.c_double()
    if lib().crd_oracle_goldbeter_steady(beta, C.byref(a), C.byref(b)) != 0:
        raise ValueError("no Goldbeter steady state")
    return a.value, b.value


def steady(model, beta):
    return fhn_steady(beta) if model == FHN else goldbeter_steady(beta)


def initial_conditions(p, wave_length, wave_width, wave_inside=0, ic_type=0, steady_state=None):
    s0, s1 = steady_state if steady_state is not None else steady(p.model, p.beta)
    ic = IC(wave_length, wave_width, wave_inside, ic_type, s0, s1)
    y = np.empty((p.nyl, p.nxl, 2), dtype=np.float64)
    if lib().crd_oracle_initial_conditions(C.byref(p), C.byref(ic), _dp(y)) != 0:
        raise ValueError("crd_oracle_initial_conditions failed")
    return y


def rhs(p, t, y, d0=1, d1=1, nthreads=1):
    """ydot = f(t, y) on the whole domain through a d0 x d1 block decomposition."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert y.shape == (p.ny, p.nx, 2)
    ydot = np.empty_like(y)
    if lib().crd_oracle_rhs_global(C.byref(p), t, _dp(y), _dp(ydot), d0, d1, nthreads) != 0:
        raise RuntimeError("crd_oracle_rhs_global failed")
    return ydot


def rhs_subdomain(p, t, y, wrecv, erecv, srecv, nrecv, nthreads=1):
    y = np.ascontiguousarray(y, dtype=np.float64)
    assert y.shape == (p.nyl, p.nxl, 2)
    ydot = np.empty_like(y)
    strips = [np.ascontiguousarray(s, dtype=np.float64) for s in (wrecv, erecv, srecv, nrecv)]
    if lib().crd_oracle_rhs_subdomain(C.byref(p), t, _dp(y), _dp(ydot), *[_dp(s) for s in strips], nthreads) != 0:
        raise RuntimeError("crd_oracle_rhs_subdomain failed")
    return ydot


def pack_edges(p, y):
    y = np.ascontiguousarray(y, dtype=np.float64)
    w, e = np.empty(2 * p.nyl), np.empty(2 * p.nyl)
    s, n = np.empty(2 * p.nxl), np.empty(2 * p.nxl)
    lib().crd_oracle_pack_edges(C.byref(p), _dp(y), _dp(w), _dp(e), _dp(s), _dp(n))
    return w, e, s, n


def rk4(p, y, t0, dt, nsteps, nthreads=1):
    """Returns y advanced by nsteps classical RK4 steps (input untouched)."""
    out = np.array(y, dtype=np.float64, order="C", copy=True)
    assert out.shape == (p.ny, p.nx, 2)
    if lib().crd_oracle_rk4(C.byref(p), _dp(out), t0, dt, nsteps, nthreads) != 0:
        raise RuntimeError("crd_oracle_rk4 failed")
    return out


def hermite(theta, h, yn, yp, fn, fp):
    """Cubic Hermite interpolant of a step [t_n, t_n + h] at t_n + theta h (ARKode's degree-3 dense output, ARK_NORMAL,
    /root/reference/src/FHNmodel_torus.cpp:423), with the coefficients formed as libcrd's launch_hermite forms them."""
    t2 = theta * theta
    t3 = t2 * theta
    h00, h10, h01, h11 = 2.0 * t3 - 3.0 * t2 + 1.0, (t3 - 2.0 * t2 + theta) * h, -2.0 * t3 + 3.0 * t2, (t3 - t2) * h
    return h00 * yn + (h10 * fn + (h01 * yp + h11 * fp))


def integrate_adaptive(p, y, t0, tout, h0, rtol=1e-5, atol=1e-10, safety=0.96, bias=1.5, growth=20.0, shrink=0.1, max_steps=200000, nthreads=1,
                       h_max=float("inf"), dense=None):
    """Error-controlled RK4(3) restated on the CPU for the tests: the step-size logic of libcrd's crd_integrate_adaptive
    around the oracle's f().  Classical RK4 propagates; k5 = f(t+h, y_new) gives the third-order embedded solution
    y + h (k1/6 + k2/3 + k3/3 + k5/6), i.e. the error estimate h (k4 - k5)/6; WRMS norm with weights 1/(rtol |y_n| + atol)
    (the tolerances of /root/reference/src/FHNmodel_torus.cpp:197-198,365).  Steps never exceed h_max (libcrd's default cap is
    its crd_stable_dt; pass that value to follow it).  Returns (y(tout), stats).
    dense: None = the last step is shortened to land on tout.  A dict (start with {}) = ARK_NORMAL: steps are never shortened,
    the value returned is the cubic Hermite interpolant of the step that passed tout, and the dict carries the integrator's
    own state (t_n, y_n, t_np1, y_np1, f_n, f_np1) to the next call, which must start at this call's tout."""
    y = np.array(y, dtype=np.float64, order="C", copy=True)
    t, h = float(t0), min(float(h0), h_max)
    st = dict(accepted=0, rejected=0, h_last=0.0, h_min=0.0, h_max=0.0, err_last=0.0, steps=[])
    after_reject = False
    n = y.size
    y_prev, t_prev = None, t
    if dense is not None and dense.get("t_out") == t0 and "y_np1" in dense:
        if tout <= dense["t_np1"]:
            hs = dense["t_np1"] - dense["t_n"]
            dense["t_out"] = tout
            st.update(h_next=h, t=tout, t_internal=dense["t_np1"])
            return hermite((tout - dense["t_n"]) / hs, hs, dense["y_n"], dense["y_np1"], dense["f_n"], dense["f_np1"]), st
        t, y = dense["t_np1"], dense["y_np1"]
    elif dense is not None:
        dense.clear()
    while t < tout:
        if st["accepted"] + st["rejected"] >= max_steps:
            raise RuntimeError("max_steps")
        hh, clipped = h, False
        if dense is None and (t + hh >= tout or tout - (t + hh) < 1e-12 * abs(tout)):
            hh, clipped = tout - t, True
        k1 = rhs(p, t, y, nthreads=nthreads)
        k2 = rhs(p, t + 0.5 * hh, y + (0.5 * hh) * k1, nthreads=nthreads)
        k3 = rhs(p, t + 0.5 * hh, y + (0.5 * hh) * k2, nthreads=nthreads)
        k4 = rhs(p, t + hh, y + hh * k3, nthreads=nthreads)
        ynew = y + (hh / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        k5 = rhs(p, t + hh, ynew, nthreads=nthreads)
        e = (hh / 6.0) * (k4 - k5) / (rtol * np.abs(y) + atol)
        err = bias * float(np.sqrt(np.sum(e * e) / n))
        st["err_last"] = err
        if not np.isfinite(err):
            eta = shrink
        elif err <= 0.0:
            eta = growth
        else:
            eta = min(growth, max(shrink, safety * err ** -0.25))
        if err <= 1.0:
            y_prev, t_prev = y, t
            t = tout if clipped else t + hh
            y = ynew
            st["accepted"] += 1
            st["steps"].append(hh)
            if after_reject:
                eta = min(eta, 1.0)
            after_reject = False
            if not clipped or st["accepted"] == 1:
                st["h_last"] = hh
                st["h_min"] = hh if st["h_min"] == 0.0 else min(st["h_min"], hh)
                st["h_max"] = max(st["h_max"], hh)
            h = min(hh * eta if not clipped else max(h, hh * eta), h_max)
        else:
            st["rejected"] += 1
            after_reject = True
            h = hh * min(eta, 0.9)
    st["h_next"] = h
    st["t"] = st["t_internal"] = t
    if dense is not None and y_prev is not None and t >= tout:
        hs = t - t_prev
        dense.update(t_out=tout, t_n=t_prev, t_np1=t, y_n=y_prev, y_np1=y, f_n=rhs(p, t_prev, y_prev, nthreads=nthreads), f_np1=rhs(p, t, y, nthreads=nthreads))
        st["t"] = tout
        return hermite((tout - t_prev) / hs, hs, y_prev, y, dense["f_n"], dense["f_np1"]), st
    return y, st
