"""Independent numpy restatement of CRDModel's RHS.  TEST INFRASTRUCTURE ONLY.

Written from the model equations and the reference's formulas, sharing no code with oracle/crd_oracle.c, so the
two restatements pin each other (the reference itself cannot be built here: PARITY UNPINNED, see crd_oracle.h).
Only tests/ may import this module.

Fields are SoA numpy arrays of shape (ny, nx): axis 0 = phi / y (index j), axis 1 = theta / x (index i).
The periodic halo is np.roll, i.e. index -1 wraps to n-1 in both directions, which is what the reference's
MPI exchange delivers at np in {1, 2, 4} (/root/reference/src/FHNmodel_torus.cpp:775-950).
"""
import math

import numpy as np

PI = 3.1415926535897932  # /root/reference/src/FHNmodel_torus.cpp:63
EPSILON = 0.36  # :68


def geometry(surface, surface_length, surface_width, nx, ny=0):
    """Returns dict(nx, ny, dx, dy, xmin, xmax, ymin, ymax, R, r).

    torus: /root/reference/src/FHNmodel_torus.cpp:73-76,188-193,233-234
    flat:  /root/reference/src/FHNmodel_flat.cpp:172-175,190-192,230-231
    """
    if surface == "torus":
        r = surface_width / (2.0 * PI)
        R = surface_length / (2.0 * PI)
        ny_ref = int(nx * (R / r))  # C double -> long truncates toward zero
        xmin, xmax, ymin, ymax = 0.0, 2.0 * PI, 0.0, 2.0 * PI
    elif surface == "flat":
        r = R = 0.0
        ny_ref = nx * int(surface_length / surface_width)
        xmin, xmax, ymin, ymax = 0.0, surface_width, 0.0, surface_length
    else:
        raise ValueError(surface)
    ny = ny if ny > 0 else ny_ref
    return dict(nx=nx, ny=ny, dx=(xmax - xmin) / (1.0 * nx - 1.0), dy=(ymax - ymin) / (1.0 * ny - 1.0),
                xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, R=R, r=r)


def diffusion(surface, g, D, u):
    """Diffusion term of the activator on the whole periodic grid.

    torus: /root/reference/src/FHNmodel_torus.cpp:535-537; flat: /root/reference/src/FHNmodel_flat.cpp:489-500.
    """
    uW, uE = np.roll(u, 1, axis=1), np.roll(u, -1, axis=1)
    uS, uN = np.roll(u, 1, axis=0), np.roll(u, -1, axis=0)
    dx, dy = g["dx"], g["dy"]
    if surface == "torus":
        R, r = g["R"], g["r"]
        theta = g["xmin"] + np.arange(g["nx"], dtype=np.float64) * dx
        rho = R + r * np.cos(theta)
        adv = (-np.sin(theta) / (r * rho))[None, :]
        return (D * (adv * (uE - uW)) / (2 * dx)
                + D * ((1 / (r * r)) * (uE - 2 * u + uW)) / (dx * dx)
                + D * ((1 / (rho * rho))[None, :] * (uN - 2 * u + uS)) / (dy * dy))
    cu1 = D / dx / dx
    cu2 = D / dy / dy
    cu3 = -2.0 * (cu1 + cu2)
    return cu1 * (uW + uE) + cu2 * (uS + uN) + cu3 * u


def beta_rows(g, beta, vary_beta, beta_min, beta_max):
    """b(j): /root/reference/src/FHNmodel_torus.cpp:623-632."""
    if vary_beta == 0:
        return np.full(g["ny"], beta)
    yy = g["ymin"] + np.arange(g["ny"], dtype=np.float64) * g["dy"]
    return beta_min + yy * (beta_max - beta_min) / (g["ymax"] - g["ymin"])


def rhs(model, surface, g, D, t, u, v, *, beta=0.0, vary_beta=0, beta_min=0.0, beta_max=0.0, t_boundary=0.0,
        just_diffusion=0):
    """(udot, vdot) of the whole domain.

    FHN kinetics /root/reference/src/FHNmodel_torus.cpp:618-664; Goldbeter kinetics
    /root/reference/src/GoldbeterModel_torus.cpp:668-721 (constants :67-78).
    """
    du = diffusion(surface, g, D, u)
    dv = np.zeros_like(v)
    if model == "goldbeter" and just_diffusion:
        return du, dv
    b = beta_rows(g, beta, vary_beta, beta_min, beta_max)[:, None]
    if model == "fhn":
        du = du + (3.0 * u - (u * u * u) - v)
        dv = dv + EPSILON * (u + b)
    elif model == "goldbeter":
        z2 = u ** 2.0
        z4 = u ** 4.0
        y2 = v ** 2.0
        v2 = 65.0 * z2 / (1.0 ** 2.0 + z2)
        v3 = 500.0 * y2 * z4 / ((2.0 ** 2.0 + y2) * (0.9 ** 4.0 + z4))
        du = du + (1.0 + 7.3 * b - v2 + v3 + 1.0 * v - 10.0 * u)
        dv = dv + (v2 - v3 - 1.0 * v)
    else:
        raise ValueError(model)
    if t < t_boundary:  # absorbing rows: global j = 0 and j = ny-1 only (:643-653)
        du[0, :] = 0.0
        dv[0, :] = 0.0
        du[-1, :] = 0.0
        dv[-1, :] = 0.0
    return du, dv


def rk4(model, surface, g, D, u, v, t0, dt, nsteps, **kw):
    """Classical RK4 with t_n = t0 + n dt (the fixed-step replacement of the ARKode loop)."""
    u = u.copy()
    v = v.copy()
    for n in range(nsteps):
        t = t0 + n * dt
        k1u, k1v = rhs(model, surface, g, D, t, u, v, **kw)
        k2u, k2v = rhs(model, surface, g, D, t + 0.5 * dt, u + (0.5 * dt) * k1u, v + (0.5 * dt) * k1v, **kw)
        k3u, k3v = rhs(model, surface, g, D, t + 0.5 * dt, u + (0.5 * dt) * k2u, v + (0.5 * dt) * k2v, **kw)
        k4u, k4v = rhs(model, surface, g, D, t + dt, u + dt * k3u, v + dt * k3v, **kw)
        u = u + (dt / 6.0) * (k1u + 2.0 * k2u + 2.0 * k3u + k4u)
        v = v + (dt / 6.0) * (k1v + 2.0 * k2v + 2.0 * k3v + k4v)
    return u, v


def goldbeter_steady(beta):
    """Fixed point: Zs = (v0 + v1 beta)/k; Ys by Newton on v2 - v3 - Y = 0 (see crd_oracle.c)."""
    Z = (1.0 + 7.3 * beta) / 10.0
    v2 = 65.0 * Z * Z / (1.0 + Z * Z)
    A = 500.0 * Z ** 4 / (0.9 ** 4 + Z ** 4)
    Y = 1.0
    for _ in range(100):
        gY = v2 - A * Y * Y / (4.0 + Y * Y) - Y
        dg = -A * 8.0 * Y / (4.0 + Y * Y) ** 2 - 1.0
        Y_new = Y - gY / dg
        if not math.isfinite(Y_new):
            raise ArithmeticError
        if abs(Y_new - Y) < 1e-15 * max(1.0, abs(Y)):
            Y = Y_new
            break
        Y = Y_new
    return Z, Y
