"""Generate the golden vectors under tests/golden/ from the C oracle.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.make_golden            # rewrites tests/golden/*.npz and MANIFEST.json

PARITY UNPINNED: these vectors come from oracle/crd_oracle.c (the CPU restatement), not from a run of the
reference, which cannot be built in this image (SUNDIALS and Boost are absent).  They pin (a) the oracle against
regressions, (b) the independent numpy restatement, and (c) the HIP path on the GPU box, where the oracle's
shared library is rebuilt from the committed C source.

Cases:
  rhs_*   single RHS evaluations on analytic inputs: FHN / Goldbeter x torus / flat, constant and varied beta,
          absorbing rows on (t < tBoundary) and off, Goldbeter justDiffusion
  rk4_*   short classical-RK4 trajectories from the reference's own initial-condition rules, crossing tBoundary
  geometry.json  ny-truncation table, slab extents, stable states
"""
import hashlib
import json
import os

import numpy as np

from . import crd_oracle as co

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

MODEL_ID = {"fhn": co.FHN, "goldbeter": co.GOLDBETER}
SURFACE_ID = {"torus": co.TORUS, "flat": co.FLAT}


def analytic_state(model, nx, ny):
    """Smooth, non-symmetric test fields; Goldbeter gets positive concentrations."""
    gi, gj = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64))
    if model == "fhn":
        u = np.sin(0.37 * gi) + np.cos(0.11 * gj) + 1e-3 * gi * gj / (nx * ny)
        v = np.cos(0.23 * gi) - np.sin(0.19 * gj)
    else:
        u = 0.8 + 0.7 * np.sin(0.37 * gi) * np.cos(0.11 * gj)      # Z in [0.1, 1.5]
        v = 1.75 + 1.25 * np.cos(0.23 * gi + 0.19 * gj)            # Y in [0.5, 3.0]
    return np.ascontiguousarray(np.stack([u, v], axis=-1))


RHS_CASES = [
    # name, model, surface, nx, L, W, ny(0=derive), kwargs
    ("rhs_fhn_torus", "fhn", "torus", 48, 40.0, 20.0, 0, dict(beta=1.25, t_boundary=5.0)),
    ("rhs_fhn_torus_varybeta", "fhn", "torus", 48, 80.0, 20.0, 0, dict(beta=1.25, vary_beta=1, beta_min=0.7, beta_max=1.7, t_boundary=5.0)),
    ("rhs_fhn_torus_ragged", "fhn", "torus", 67, 80.0, 20.0, 53, dict(beta=1.25, t_boundary=5.0)),
    ("rhs_fhn_flat", "fhn", "flat", 64, 20.0, 20.0, 0, dict(beta=1.25, t_boundary=5.0)),
    ("rhs_fhn_flat_varybeta", "fhn", "flat", 40, 40.0, 20.0, 0, dict(beta=1.25, vary_beta=1, beta_min=0.7, beta_max=1.7, t_boundary=5.0)),
    ("rhs_goldbeter_torus", "goldbeter", "torus", 48, 40.0, 20.0, 0, dict(beta=0.4, t_boundary=5.0)),
    ("rhs_goldbeter_flat", "goldbeter", "flat", 48, 40.0, 20.0, 0, dict(beta=0.4, vary_beta=1, beta_min=0.0, beta_max=1.0, t_boundary=5.0)),
    ("rhs_goldbeter_torus_justdiffusion", "goldbeter", "torus", 48, 40.0, 20.0, 0, dict(beta=0.4, just_diffusion=1, t_boundary=5.0)),
]

RK4_CASES = [
    # name, model, surface, nx, L, W, ny, params, ic kwargs, dt, snapshots (step counts)
    ("rk4_fhn_torus_outside", "fhn", "torus", 32, 80.0, 20.0, 0, dict(beta=1.25, t_boundary=5.0),
     dict(wave_length=0.1, wave_width=0.5, wave_inside=0), 0.05, [50, 100, 200]),
    ("rk4_fhn_torus_inside", "fhn", "torus", 32, 80.0, 20.0, 0, dict(beta=1.25, t_boundary=5.0),
     dict(wave_length=0.1, wave_width=0.5, wave_inside=1), 0.05, [50, 100, 200]),
    ("rk4_fhn_flat", "fhn", "flat", 32, 80.0, 20.0, 0, dict(beta=1.25, t_boundary=2.0),
     dict(wave_length=0.1, wave_width=0.5), 0.05, [60, 120]),
    ("rk4_goldbeter_torus", "goldbeter", "torus", 32, 80.0, 20.0, 0, dict(beta=0.4, t_boundary=0.5),
     dict(wave_length=0.2, wave_width=0.5, wave_inside=1), 0.002, [250, 500]),
]

DIFFUSION = 0.12


def problem(model, surface, nx, L, W, ny, kw):
    return co.make_problem(MODEL_ID[model], SURFACE_ID[surface], nx, L, W, DIFFUSION, kw.get("beta", 0.0), ny=ny,
                           beta_min=kw.get("beta_min", 0.0), beta_max=kw.get("beta_max", 0.0), vary_beta=kw.get("vary_beta", 0),
                           just_diffusion=kw.get("just_diffusion", 0), t_boundary=kw.get("t_boundary", 0.0))


def meta_of(model, surface, nx, L, W, ny, kw, p, extra=None):
    m = dict(model=model, surface=surface, nx=nx, ny=int(p.ny), ny_override=ny, surface_length=L, surface_width=W,
             diffusion=DIFFUSION, beta=kw.get("beta", 0.0), beta_min=kw.get("beta_min", 0.0), beta_max=kw.get("beta_max", 0.0),
             vary_beta=kw.get("vary_beta", 0), just_diffusion=kw.get("just_diffusion", 0), t_boundary=kw.get("t_boundary", 0.0))
    if extra:
        m.update(extra)
    return m


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {}

    def save(name, meta, **arrays):
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, meta=json.dumps(meta), **arrays)
        h = hashlib.sha256()
        for k in sorted(arrays):
            h.update(np.ascontiguousarray(arrays[k]).tobytes())
        manifest[name] = dict(meta=meta, sha256_arrays=h.hexdigest(), arrays={k: list(v.shape) for k, v in arrays.items()})

    for name, model, surface, nx, L, W, ny, kw in RHS_CASES:
        p = problem(model, surface, nx, L, W, ny, kw)
        y = analytic_state(model, p.nx, p.ny)
        t_on, t_off = 0.0, 50.0  # t < tBoundary: absorbing rows active; t >= tBoundary: off
        save(name, meta_of(model, surface, nx, L, W, ny, kw, p, dict(t_absorbing=t_on, t_free=t_off)),
             y=y, ydot_absorbing=co.rhs(p, t_on, y), ydot_free=co.rhs(p, t_off, y))

    for name, model, surface, nx, L, W, ny, kw, ickw, dt, snaps in RK4_CASES:
        p = problem(model, surface, nx, L, W, ny, kw)
        s0, s1 = co.steady(MODEL_ID[model], kw["beta"])
        y0 = co.initial_conditions(p, ickw["wave_length"], ickw["wave_width"], ickw.get("wave_inside", 0), 0, (s0, s1))
        arrays = dict(y0=y0)
        y, done = y0, 0
        for n in snaps:
            y = co.rk4(p, y, done * dt, dt, n - done)
            done = n
            arrays["y_%d" % n] = y
        save(name, meta_of(model, surface, nx, L, W, ny, kw, p, dict(ic=ickw, dt=dt, snapshots=snaps, steady=[s0, s1])), **arrays)

    # geometry table (G8): ny truncation, slab extents, stable states
    geo = {"ny": [], "slabs": [], "steady": []}
    for surface, L, W, nx in [("torus", 100.0, 20.0, 100), ("torus", 100.0, 20.0, 400), ("torus", 80.0, 20.0, 400), ("torus", 40.0, 20.0, 200),
                              ("torus", 80.0, 20.0, 101), ("flat", 80.0, 20.0, 400), ("flat", 90.0, 20.0, 100), ("flat", 20.0, 20.0, 256)]:
        p = co.make_problem(co.FHN, SURFACE_ID[surface], nx, L, W, DIFFUSION, 1.25)
        geo["ny"].append(dict(surface=surface, L=L, W=W, nx=nx, ny=int(p.ny), dx=p.dx, dy=p.dy, R=p.R, r=p.r))
    for ny, G in [(1600, 1), (1600, 2), (1600, 8), (499, 4), (499, 8), (8192, 8), (53, 3)]:
        ext = []
        for gidx in range(G):
            q = co.subproblem(co.make_problem(co.FHN, co.FLAT, 8, 8.0 * 1, 8.0, DIFFUSION, 1.0, ny=ny), 1, G, 0, gidx)
            ext.append([int(q.js), int(q.je)])
        geo["slabs"].append(dict(ny=ny, n_slabs=G, extents=ext))
    for beta in (0.14, 0.4, 0.6, 1.0):
        geo["steady"].append(dict(model="goldbeter", beta=beta, state=list(co.goldbeter_steady(beta))))
    for beta in (0.7, 1.25):
        geo["steady"].append(dict(model="fhn", beta=beta, state=list(co.fhn_steady(beta))))
    with open(os.path.join(GOLDEN, "geometry.json"), "w") as f:
        json.dump(geo, f, indent=1)

    with open(os.path.join(GOLDEN, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(GOLDEN, f)) for f in os.listdir(GOLDEN))
    print("wrote %d cases, %.1f KiB under %s" % (len(manifest), total / 1024.0, GOLDEN))


if __name__ == "__main__":
    main()
