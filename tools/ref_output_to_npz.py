#!/usr/bin/env python3
"""Pack the output files of one run of the UNMODIFIED reference program (tools/make_reference_fixtures.md) into a fixture:

    tools/ref_output_to_npz.py --ini run.ini --dir rundir --model fhn|goldbeter --surface torus|flat --tag NAME --np N --out tests/golden/ref_NAME_npN.npz

Stored: the run configuration as libcrd's ini reader sees it, the subdomain rectangles, every output row of both fields stitched to
the whole grid (text `%.16e` round-trips doubles), and a free-text provenance note."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crdmodel_amd as crd  # noqa: E402  (host-side helpers only: no GPU needed)
from crdmodel_amd import post  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ini", required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--model", required=True, choices=["fhn", "goldbeter"])
    ap.add_argument("--surface", required=True, choices=["torus", "flat"])
    ap.add_argument("--tag", required=True)
    ap.add_argument("--np", type=int, required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    cfg = crd.load_ini(a.ini, a.model, a.surface)
    run = post.load_run(a.dir, a.model, a.surface, include_all_vars=True)
    if len(run.subdomains) != a.np:
        raise SystemExit("%d subdomain files, expected %d" % (len(run.subdomains), a.np))
    names = list(run.fields)
    np.savez_compressed(a.out, tag=a.tag, model=a.model, surface=a.surface, nprocs=a.np, note=a.note, ini_text=open(a.ini).read(),
                        config=np.frombuffer(bytes(cfg), dtype=np.uint8), config_size=C.sizeof(cfg), nx=run.nx, ny=run.ny, t_final=run.t_final,
                        subdomains=run.subdomains, var0=run.fields[names[0]], var1=run.fields[names[1]] if len(names) > 1 else np.zeros(0))
    print("wrote %s: %d x %d, %d output rows, fields %s" % (a.out, run.nx, run.ny, run.fields[names[0]].shape[0], names))


if __name__ == "__main__":
    main()
