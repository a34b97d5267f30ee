"""crdmodel_amd.post: the Python-3 counterparts of the reference's loaders, frame plots and torus mapping, checked on files
the library's own writer produced (CPU only)."""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import crdmodel_amd as crd
from crdmodel_amd import post
from conftest import GOLDEN

INI = os.path.join(GOLDEN, "ini")


def write_run(directory, cfg, frames, n_slabs):
    g = crd.grid_of(cfg.params)
    for k in range(n_slabs):
        js, je = crd.slab_extents(g.ny, k, n_slabs)
        with crd.Writer(cfg, directory, k, n_slabs) as w:
            for f in frames:
                w.write_row(f[js:je + 1])
    return g


@pytest.mark.parametrize("n_slabs", [1, 3])
def test_load_run_stitches_subdomains(tmp_path, n_slabs):
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    rng = np.random.default_rng(5)
    g = crd.grid_of(cfg.params)
    frames = [rng.standard_normal((g.ny, g.nx, 2)) for _ in range(4)]
    write_run(tmp_path, cfg, frames, n_slabs)
    run = post.load_run(tmp_path, "fhn", "torus", include_all_vars=bool(cfg.include_all_vars))
    assert (run.nx, run.ny, run.nt) == (g.nx, g.ny, 4) and run.subdomains.shape == (n_slabs, 4)
    assert run.t_final == cfg.t_final and run.xmin == 0.0 and run.xmax == pytest.approx(2 * math.pi, abs=1e-6)
    for k, f in enumerate(frames):
        assert np.array_equal(run.activator[k], f[..., 0])
        if cfg.include_all_vars:
            assert np.array_equal(run.fields["v"][k], f[..., 1])
    assert post.frame_time(run, 2) == pytest.approx(2 / 4 * cfg.t_final)  # the plot scripts' label: tstep / nt * tFinal


def test_load_run_rejects_inconsistent_files(tmp_path):
    with pytest.raises(FileNotFoundError):
        post.load_run(tmp_path, "fhn", "torus")
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    g = crd.grid_of(cfg.params)
    frames = [np.zeros((g.ny, g.nx, 2))] * 2
    write_run(tmp_path, cfg, frames, 2)
    with open(tmp_path / "FHNmodel_torus_u.001.txt", "a") as f:  # one more output row in subdomain 1 than in subdomain 0
        js, je = crd.slab_extents(g.ny, 1, 2)
        f.write(" 0.0" * ((je - js + 1) * g.nx) + "\n")
    with pytest.raises(ValueError, match="time steps"):
        post.load_run(tmp_path, "fhn", "torus")


def test_cell_index_is_the_reference_rule():
    """A scalar re-derivation of XYZtoRC (util/FHNmodel/MapOutputToTorus.py:16-35) on hand-picked points."""
    R, r, ny, nx = 80 / (2 * math.pi), 20 / (2 * math.pi), 1600, 400
    pts = []
    for phi, theta in [(0.3, 0.2), (2.0, 1.7), (4.0, 3.5), (6.0, 5.9), (1.0, math.pi / 2 - 1e-3), (5.5, 3 * math.pi / 2 + 1e-3)]:
        rho = R + r * math.cos(theta)
        pts.append((rho * math.cos(phi), r * math.sin(theta), rho * math.sin(phi), phi, theta))
    a = np.array(pts)
    phi, theta, row, col = post.cell_index(a[:, :3], ny, nx, r, R)
    assert np.allclose(phi, a[:, 3], atol=1e-12) and np.allclose(theta, a[:, 4], atol=1e-9)
    for k, (x, y, z, ph, th) in enumerate(pts):
        p = math.atan2(z, x) % (2 * math.pi)
        t = (math.asin(y / r) if math.sqrt(x * x + z * z) > R else math.pi - math.asin(y / r)) % (2 * math.pi)
        assert (row[k], col[k]) == (int(p / (2 * math.pi) * (ny - 1)), int(t / (2 * math.pi) * (nx - 1)))


def test_torus_mesh_geometry():
    R, r = 3.0, 1.0
    pts, quads = post.torus_mesh(R, r, 16, 48)
    assert pts.shape == (16 * 48, 3) and quads.shape == (16 * 48, 4) and quads.min() == 0 and quads.max() == len(pts) - 1
    rho = np.sqrt(pts[:, 0] ** 2 + pts[:, 2] ** 2)
    assert np.allclose((rho - R) ** 2 + pts[:, 1] ** 2, r * r)  # every vertex lies on the torus
    assert np.all(np.bincount(quads.ravel()) == 4)  # closed surface: every vertex belongs to four quads
    edge = np.linalg.norm(pts[quads[:, 1]] - pts[quads[:, 0]], axis=1)
    assert edge.max() < 2 * math.pi * r / 16 * 1.01  # theta edges of a cell are short chords of the minor circle


def test_map_to_torus_writes_vtp_and_pvd(tmp_path):
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    g = crd.grid_of(cfg.params)
    jj, ii = np.meshgrid(np.arange(g.ny), np.arange(g.nx), indexing="ij")
    frames = [np.stack([1000.0 * t + jj + ii / 1000.0, -jj.astype(float)], axis=-1) for t in range(2)]  # value encodes (t, row, col)
    write_run(tmp_path, cfg, frames, 2)
    run = post.load_run(tmp_path, "fhn", "torus", include_all_vars=True) if cfg.include_all_vars else post.load_run(tmp_path, "fhn", "torus")
    p = cfg.params
    steps = post.map_to_torus(run, p.surface_length, p.surface_width, out_dir=tmp_path / "FHNstep", pvd=tmp_path / "FHNtimeSteps.pvd", n_theta=12,
                              vary_beta=True)
    assert [os.path.basename(s[1]) for s in steps] == ["FHNstep_000.vtp", "FHNstep_001.vtp"]
    n_phi = int(12 * (p.surface_length / p.surface_width))
    root = ET.parse(steps[1][1]).getroot()
    piece = root.find("PolyData/Piece")
    assert int(piece.get("NumberOfPoints")) == 12 * n_phi and int(piece.get("NumberOfPolys")) == 12 * n_phi
    arrays = {d.get("Name"): np.array(d.text.split(), dtype=float) for d in piece.find("CellData")}
    assert set(arrays) >= {"Activator", "Hopf Bifurcations"}
    # every cell carries the value of the result entry its centre maps to
    pts, quads = post.torus_mesh(p.surface_length / (2 * math.pi), p.surface_width / (2 * math.pi), 12, n_phi)
    phi, _, row, col = post.cell_index(pts[quads].mean(axis=1), g.ny, g.nx, p.surface_width / (2 * math.pi), p.surface_length / (2 * math.pi))
    assert np.array_equal(arrays["Activator"], 1000.0 + row + col / 1000.0)
    hopf = post.hopf_position(*post.FHN_MAP_BETA_RANGE)
    assert np.array_equal(arrays["Hopf Bifurcations"] == 1.0, np.abs(phi - hopf) < 0.01)
    pvd = ET.parse(tmp_path / "FHNtimeSteps.pvd").getroot()
    sets = pvd.findall("Collection/DataSet")
    assert [s.get("timestep") for s in sets] == ["%.1f" % post.frame_time(run, k) for k in range(2)]


def test_plot_frames_writes_pngs(tmp_path):
    pytest.importorskip("matplotlib")
    cfg = crd.load_ini(os.path.join(INI, "small_run.ini"), "fhn", "torus")
    g = crd.grid_of(cfg.params)
    frames = [np.full((g.ny, g.nx, 2), float(t + 1)) for t in range(2)]
    write_run(tmp_path, cfg, frames, 1)
    run = post.load_run(tmp_path, "fhn", "torus")
    files = post.plot_frames(run, tmp_path / "png", beta="1.25", dpi=40)
    assert [os.path.basename(f) for f in files] == ["FHNmodel_torus_Z.beta1.25.000.png", "FHNmodel_torus_Z.beta1.25.001.png"]
    assert all(open(f, "rb").read(8) == b"\x89PNG\r\n\x1a\n" for f in files)
    files = post.plot_frames(run, tmp_path / "png", vary_beta=True, beta_min=0.7, beta_max=1.7, dpi=40)
    assert os.path.basename(files[1]) == "FHNmodel_torus_Z.varyBeta_linear001.png"


def test_command_line(tmp_path, capsys):
    pytest.importorskip("matplotlib")
    ini = os.path.join(INI, "small_run.ini")
    cfg = crd.load_ini(ini, "fhn", "torus")
    g = crd.grid_of(cfg.params)
    write_run(tmp_path, cfg, [np.ones((g.ny, g.nx, 2)), np.zeros((g.ny, g.nx, 2))], 2)
    post.main(["plot", ini, "--dir", str(tmp_path)])
    post.main(["map", ini, "--dir", str(tmp_path), "--mesh", "8"])
    out = capsys.readouterr().out
    assert "wrote 2 frames" in out and "wrote 2 time steps" in out
    assert os.path.exists(tmp_path / "png" / "FHNmodel_torus_Z.beta1.25.001.png")
    assert os.path.exists(tmp_path / "FHNstep" / "FHNstep_001.vtp") and os.path.exists(tmp_path / "FHNtimeSteps.pvd")
    with pytest.raises(SystemExit):
        post.main(["map", ini, "--dir", str(tmp_path), "--surface", "flat"])
