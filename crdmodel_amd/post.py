"""Python-3 counterparts of the reference's post-processing utilities, working on the per-subdomain text files the driver
(crdmodel_amd/bin/crd_run and its aliases) or the reference itself writes.  Host-side only: numpy, plus matplotlib for the
frame plots; no VTK, configobj or lxml needed (the geometry and the VTK XML files are produced here).

What each piece stands in for (behaviour restated, nothing shared with those scripts):
  load_run        the loader block of util/FHNmodel/plot_FHNmodel_torus.py:26-87 (= the flat / Goldbeter plot scripts and
                  MapOutputToTorus.py:66-134): probe `<prefix>_subdomain.NNN.txt` until one is missing, read each header
                  `nx ny is ie js je xmin xmax tfinal`, paste every row of `<prefix>_<var>.NNN.txt` at [js:je+1, is:ie+1]
  frame_time      the time label of output row k, `k / nt * tFinal` (plot_FHNmodel_torus.py:125)
  plot_frames     the imshow frames `png/<prefix>_Z.beta<beta>.NNN.png` (plot_FHNmodel_torus.py:94-131), dashed line at the
                  Hopf position for varyBeta runs (`:90-92,121-123`)
  torus_mesh      the geometry util/GenTorus.py:21-52 obtains from vtkSuperquadricSource: a torus of major radius R and minor
                  radius r around the y axis, n_theta x n_phi cells (quads here, no triangulation)
  cell_index      XYZtoRC of MapOutputToTorus.py:16-35: cell centre (x, y, z) -> (phi, theta) -> (row, column) of the results
  map_to_torus    MapOutputToTorus.py:137-196: per output row a `.vtp` with cell arrays Activator [/ Inhibitor / Hopf
                  Bifurcations], plus the `.pvd` collection (`:198-219`)

Command line:  python -m crdmodel_amd.post plot|map <ini> [--model fhn|goldbeter] [--surface torus|flat] [--dir run_dir]
"""
import argparse
import math
import os
from dataclasses import dataclass, field
from xml.sax.saxutils import quoteattr

import numpy as np

_VARS = {"FHNmodel": ("u", "v"), "GoldbeterModel": ("Z", "Y")}
_MODEL_PREFIX = {"fhn": "FHNmodel", "goldbeter": "GoldbeterModel"}
# constants of the reference's mapping scripts: the beta range they assume for the FHN Hopf marker
# (util/FHNmodel/MapOutputToTorus.py:60-64) and the two Goldbeter Hopf positions (util/GoldbeterModel/MapOutputToTorus.py:60-63)
FHN_MAP_BETA_RANGE = (0.7, 1.7)
GOLDBETER_HOPF_FRACTIONS = (0.289, 0.774)


def prefix_of(model, surface):
    return "%s_%s" % (_MODEL_PREFIX[model], surface)


@dataclass
class Run:
    """A whole run stitched back together: fields[name] has shape (nt, ny, nx), row-major phi (j) then theta (i)."""
    prefix: str
    nx: int
    ny: int
    xmin: float
    xmax: float
    t_final: float
    subdomains: np.ndarray  # (nprocs, 4): is, ie, js, je
    fields: dict = field(default_factory=dict)

    @property
    def nt(self):
        return next(iter(self.fields.values())).shape[0]

    @property
    def activator(self):
        return self.fields[_VARS[self.prefix.split("_")[0]][0]]


def load_run(directory, model, surface, include_all_vars=False):
    """Stitch the subdomain files of one run.  Raises ValueError on incompatible headers or row counts."""
    prefix = prefix_of(model, surface)
    names = _VARS[_MODEL_PREFIX[model]]
    nprocs = 0
    while os.path.exists(os.path.join(directory, "%s_subdomain.%03d.txt" % (prefix, nprocs))):
        nprocs += 1
    if nprocs == 0:
        raise FileNotFoundError("no %s_subdomain.000.txt in %s" % (prefix, directory))
    run = None
    for k in range(nprocs):
        head = np.loadtxt(os.path.join(directory, "%s_subdomain.%03d.txt" % (prefix, k)), dtype=np.float64)
        if head.shape != (9,):
            raise ValueError("subdomain header %d does not have 9 fields" % k)
        nx, ny = int(head[0]), int(head[1])
        i0, i1, j0, j1 = (int(v) for v in head[2:6])
        if run is None:
            run = Run(prefix, nx, ny, float(head[6]), float(head[7]), float(head[8]), np.zeros((nprocs, 4), dtype=np.int64))
        elif (nx, ny) != (run.nx, run.ny):
            raise ValueError("subdomain files incompatible (clean up and re-run)")
        run.subdomains[k] = (i0, i1, j0, j1)
        nxl, nyl = i1 - i0 + 1, j1 - j0 + 1
        for name in names[:2 if include_all_vars else 1]:
            # the driver's binary side-channel (`crd_run --binary`), when present, holds the same frames as the text file
            npy = os.path.join(directory, "%s_%s.%03d.npy" % (prefix, name, k))
            if os.path.exists(npy):
                frames = np.load(npy)
                if frames.ndim != 3 or frames.shape[1:] != (nyl, nxl):
                    raise ValueError("subdomain %d: %s holds frames of shape %r, expected %r" % (k, npy, frames.shape[1:], (nyl, nxl)))
                rows = frames.reshape(frames.shape[0], nyl * nxl).astype(np.float64)
            else:
                rows = np.loadtxt(os.path.join(directory, "%s_%s.%03d.txt" % (prefix, name, k)), dtype=np.float64, ndmin=2)
            if rows.shape[1] != nxl * nyl:
                raise ValueError("subdomain %d: rows of %s have %d values, expected %d" % (k, name, rows.shape[1], nxl * nyl))
            if name not in run.fields:
                run.fields[name] = np.zeros((rows.shape[0], ny, nx))
            if rows.shape[0] != run.fields[name].shape[0]:
                raise ValueError("subdomain %d has an incorrect number of time steps" % k)
            run.fields[name][:, j0:j1 + 1, i0:i1 + 1] = rows.reshape(rows.shape[0], nyl, nxl)
    return run


def frame_time(run, k, t_final=None):
    return (k / run.nt) * (run.t_final if t_final is None else t_final)


def hopf_position(beta_min, beta_max):
    """phi at which beta(phi) = beta_min + (beta_max - beta_min) phi / 2 pi crosses 1 (the FHN Hopf bifurcation)."""
    return (1.0 - beta_min) * 2.0 * math.pi / (beta_max - beta_min)


def plot_frames(run, out_dir="png", beta="", vary_beta=False, beta_min=0.0, beta_max=0.0, surface_length=None, var=None, dpi=150):
    """One imshow frame per output row, colour range 0.9 min .. 1.1 max over the whole run; returns the file names."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    data = run.activator if var is None else run.fields[var]
    vmin, vmax = 0.9 * data.min(), 1.1 * data.max()
    torus = run.prefix.endswith("torus")
    ymax = 2.0 * math.pi if torus else (surface_length if surface_length is not None else float(run.ny - 1))
    os.makedirs(out_dir, exist_ok=True)
    files = []
    for k in range(run.nt):
        stem = "%s_Z.varyBeta_linear%03d.png" % (run.prefix, k) if vary_beta else "%s_Z.beta%s.%03d.png" % (run.prefix, beta, k)
        fig, ax = plt.subplots()
        img = ax.imshow(data[k], extent=[run.xmin, run.xmax, 0.0, ymax], cmap="jet", aspect="auto", vmin=vmin, vmax=vmax, origin="lower")
        ax.set_xlabel("theta" if torus else "x")
        ax.set_ylabel("phi" if torus else "y")
        fig.colorbar(img)
        if vary_beta and torus and beta_max != beta_min:
            ax.axhline(y=hopf_position(beta_min, beta_max), color="r", linewidth=1, linestyle="dashed")
        ax.set_title("%s: %s at t = %.1f, mesh = %dx%d" % ("Torus" if torus else "Flat", "u" if var is None else var, frame_time(run, k), run.nx, run.ny))
        path = os.path.join(out_dir, stem)
        fig.savefig(path, dpi=dpi)
        plt.close(fig)
        files.append(path)
    return files


# ---- torus geometry and mapping -------------------------------------------------------------------------------------

def torus_mesh(R, r, n_theta, n_phi):
    """Quad mesh of the torus ((R + r cos theta) cos phi, r sin theta, (R + r cos theta) sin phi): points (n_phi*n_theta, 3),
    quads (n_phi*n_theta, 4), both periodic.  phi = atan2(z, x) and theta is measured from the outer equator, the convention
    the reference's XYZtoRC inverts."""
    th = 2.0 * math.pi * np.arange(n_theta) / n_theta
    ph = 2.0 * math.pi * np.arange(n_phi) / n_phi
    T, P = np.meshgrid(th, ph)  # (n_phi, n_theta)
    rho = R + r * np.cos(T)
    pts = np.stack([rho * np.cos(P), r * np.sin(T), rho * np.sin(P)], axis=-1).reshape(-1, 3)
    j, i = np.meshgrid(np.arange(n_phi), np.arange(n_theta), indexing="ij")
    jn, inx = (j + 1) % n_phi, (i + 1) % n_theta
    quads = np.stack([j * n_theta + i, j * n_theta + inx, jn * n_theta + inx, jn * n_theta + i], axis=-1).reshape(-1, 4)
    return pts, quads


def cell_index(centres, ny, nx, r, R):
    """(phi, theta, row, col) of each cell centre, by the reference's rule: phi = atan2(z, x) mod 2 pi; theta = asin(y / r)
    on the outer half (sqrt(x^2 + z^2) > R), pi - asin(y / r) on the inner half, mod 2 pi; row = int(phi / 2 pi (ny - 1)),
    col = int(theta / 2 pi (nx - 1))."""
    x, y, z = centres[:, 0], centres[:, 1], centres[:, 2]
    two_pi = 2.0 * math.pi
    phi = np.mod(np.arctan2(z, x), two_pi)
    s = np.arcsin(np.clip(y / r, -1.0, 1.0))
    theta = np.mod(np.where(np.sqrt(x * x + z * z) > R, s, math.pi - s), two_pi)
    row = (phi / two_pi * (ny - 1)).astype(np.int64)
    col = (theta / two_pi * (nx - 1)).astype(np.int64)
    return phi, theta, row, col


def _data_array(f, dtype, name, values, components=None):
    attrs = 'type="%s" format="ascii"' % dtype
    if name:
        attrs += " Name=%s" % quoteattr(name)
    if components:
        attrs += ' NumberOfComponents="%d"' % components
    f.write("    <DataArray %s>\n" % attrs)
    flat = np.asarray(values).reshape(-1)
    fmt = "%d" if flat.dtype.kind in "iu" else "%.17g"
    for a in range(0, flat.size, 12):
        f.write("     " + " ".join(fmt % v for v in flat[a:a + 12]) + "\n")
    f.write("    </DataArray>\n")


def write_vtp(path, points, quads, cell_arrays):
    """VTK XML PolyData (ASCII) with one Float64 cell array per entry of `cell_arrays` (the first one is the active scalar)."""
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="PolyData" version="0.1" byte_order="LittleEndian">\n <PolyData>\n')
        f.write('  <Piece NumberOfPoints="%d" NumberOfVerts="0" NumberOfLines="0" NumberOfStrips="0" NumberOfPolys="%d">\n' % (len(points), len(quads)))
        f.write("   <Points>\n")
        _data_array(f, "Float64", None, points, components=3)
        f.write("   </Points>\n   <Polys>\n")
        _data_array(f, "Int64", "connectivity", quads.astype(np.int64))
        _data_array(f, "Int64", "offsets", 4 * np.arange(1, len(quads) + 1, dtype=np.int64))
        f.write("   </Polys>\n")
        names = list(cell_arrays)
        f.write("   <CellData Scalars=%s>\n" % quoteattr(names[0]) if names else "   <CellData>\n")
        for name in names:
            _data_array(f, "Float64", name, np.asarray(cell_arrays[name], dtype=np.float64))
        f.write("   </CellData>\n  </Piece>\n </PolyData>\n</VTKFile>\n")


def write_pvd(path, steps):
    """ParaView collection: steps = [(time, file), ...]; times are written with one decimal like the reference's."""
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n <Collection>\n')
        for t, name in steps:
            f.write('  <DataSet timestep="%.1f" group="" part="0" file=%s/>\n' % (t, quoteattr(name)))
        f.write(" </Collection>\n</VTKFile>\n")


def map_to_torus(run, surface_length, surface_width, out_dir=None, pvd=None, n_theta=None, vary_beta=False, hopf_width=0.01):
    """Paint every output row of a torus run onto the torus surface.  Returns [(time, vtp path), ...]."""
    model = run.prefix.split("_")[0]
    short = "FHN" if model == "FHNmodel" else "Goldbeter"
    out_dir = out_dir or "%sstep" % short
    pvd = pvd or "%stimeSteps.pvd" % short
    r, R = surface_width / (2.0 * math.pi), surface_length / (2.0 * math.pi)
    n_theta = n_theta or run.nx
    n_phi = int(n_theta * (R / r))
    pts, quads = torus_mesh(R, r, n_theta, n_phi)
    centres = pts[quads].mean(axis=1)
    phi, _, row, col = cell_index(centres, run.ny, run.nx, r, R)
    os.makedirs(out_dir, exist_ok=True)
    names = _VARS[model]
    if model == "FHNmodel":
        marks = [hopf_position(*FHN_MAP_BETA_RANGE)]
    else:
        marks = [f * 2.0 * math.pi for f in GOLDBETER_HOPF_FRACTIONS]
    steps = []
    for k in range(run.nt):
        arrays = {"Activator": run.fields[names[0]][k, row, col]}
        if vary_beta:
            arrays["Hopf Bifurcations"] = np.where(np.min([np.abs(phi - m) for m in marks], axis=0) < hopf_width, 1.0, 0.0)
        if names[1] in run.fields:
            arrays["Inhibitor"] = run.fields[names[1]][k, row, col]
        path = os.path.join(out_dir, "%sstep_%03d.vtp" % (short, k))
        write_vtp(path, pts, quads, arrays)
        steps.append((frame_time(run, k), path))
    write_pvd(pvd, steps)
    return steps


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("what", choices=["plot", "map"])
    ap.add_argument("ini")
    ap.add_argument("--model", default="fhn", choices=["fhn", "goldbeter"])
    ap.add_argument("--surface", default="torus", choices=["torus", "flat"])
    ap.add_argument("--dir", default=".", help="directory holding the subdomain files")
    ap.add_argument("--mesh", type=int, default=0, help="map: theta cells of the torus mesh (default: the run's nx)")
    a = ap.parse_args(argv)
    if a.what == "map" and a.surface != "torus":
        raise SystemExit("map needs a torus run")
    from . import solver  # the ini reader is libcrd's (crd_config_load_ini), so both sides parse the file identically

    cfg = solver.load_ini(a.ini, a.model, a.surface)
    p = cfg.params
    run = load_run(a.dir, a.model, a.surface, include_all_vars=bool(cfg.include_all_vars))
    if a.what == "plot":
        files = plot_frames(run, os.path.join(a.dir, "png"), beta="%g" % p.beta, vary_beta=bool(p.vary_beta), beta_min=p.beta_min, beta_max=p.beta_max,
                            surface_length=p.surface_length)
        print("wrote %d frames to %s" % (len(files), os.path.join(a.dir, "png")))
    else:
        short = "FHN" if a.model == "fhn" else "Goldbeter"
        steps = map_to_torus(run, p.surface_length, p.surface_width, out_dir=os.path.join(a.dir, "%sstep" % short),
                             pvd=os.path.join(a.dir, "%stimeSteps.pvd" % short), n_theta=a.mesh or None, vary_beta=bool(p.vary_beta))
        print("wrote %d time steps" % len(steps))


if __name__ == "__main__":
    main()
