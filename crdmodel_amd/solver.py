"""Host-side mirror of the reference's interface for the RHS path, on top of the libcrd C ABI.

Names follow the reference (/root/reference/src/FHNmodel_torus.cpp): a `Slab` is one subdomain (UserData) living
on one GPU; `Slab.f(t, y)` is the ARKRhsFn `f(t, y, ydot, user_data)` (:504); `Slab.step_rk4` replaces the
`ARKode(...)` call (:423).  Vectors are numpy arrays in the reference's layout, shape (nyl, nx, 2) = AoS [var0, var1],
theta fastest.  All arithmetic happens in the HIP kernels; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import (MODELS, SURFACES, STEPPERS, PRECISION_F32, PRECISION_F64, CrdError, Grid, Params,  # noqa: F401
                    RunConfig, check, lib)


def make_params(model, surface, nx, surface_length, surface_width, diffusion, beta, *, ny=0, beta_min=0.0, beta_max=0.0,
                vary_beta=0, just_diffusion=0, t_boundary=0.0, precision="f64"):
    p = Params()
    p.model = MODELS[model] if isinstance(model, str) else model
    p.surface = SURFACES[surface] if isinstance(surface, str) else surface
    p.nx, p.ny = nx, ny
    p.surface_length, p.surface_width = surface_length, surface_width
    p.diffusion, p.beta, p.beta_min, p.beta_max = diffusion, beta, beta_min, beta_max
    p.vary_beta, p.just_diffusion = vary_beta, just_diffusion
    p.t_boundary = t_boundary
    p.precision = {"f64": PRECISION_F64, "f32": PRECISION_F32}[precision] if isinstance(precision, str) else precision
    return p


def grid_of(params):
    g = Grid()
    check(lib().crd_grid_from_params(C.byref(params), C.byref(g)), "crd_grid_from_params")
    return g


def slab_extents(ny, slab, n_slabs):
    js, je = C.c_int64(), C.c_int64()
    check(lib().crd_slab_extents(ny, slab, n_slabs, C.byref(js), C.byref(je)), "crd_slab_extents")
    return js.value, je.value


def dims_create(nprocs):
    """MPI_Dims_create(nprocs, 2): the reference's process grid (d0 theta-blocks x d1 phi-blocks)."""
    d0, d1 = C.c_int(), C.c_int()
    check(lib().crd_dims_create(nprocs, C.byref(d0), C.byref(d1)), "crd_dims_create")
    return d0.value, d1.value


def block_extents(nx, ny, c0, d0, c1, d1):
    """(is, ie, js, je) of block (c0, c1) of a d0 x d1 decomposition (SetupDecomp)."""
    v = [C.c_int64() for _ in range(4)]
    check(lib().crd_block_extents(nx, ny, c0, d0, c1, d1, *[C.byref(x) for x in v]), "crd_block_extents")
    return tuple(x.value for x in v)


def halo_plan(slab, n_slabs, nyl, depth=1):
    """The four ordered point-to-point operations of one halo exchange: [(is_send, peer, row_begin, row_count)]."""
    ops = (capi.HaloOp * 4)()
    check(lib().crd_halo_plan(slab, n_slabs, nyl, depth, ops), "crd_halo_plan")
    return [(bool(o.is_send), o.peer, o.row_begin, o.row_count) for o in ops]


def cycle_vote(pos):
    """What a rank standing at position `pos` of the exchange cycle (-1: ghost rows not trusted) contributes to the agreement."""
    v = (C.c_double * 2)()
    check(lib().crd_cycle_vote(pos, v), "crd_cycle_vote")
    return [v[0], v[1]]


def cycle_agreed(reduced):
    """The ring's common cycle position from the element-wise MIN of the ranks' votes, or -1 (start with an exchange)."""
    return lib().crd_cycle_agreed((C.c_double * 2)(*reduced))


def steady_state(model, beta):
    a, b = C.c_double(), C.c_double()
    m = MODELS[model] if isinstance(model, str) else model
    check(lib().crd_steady_state(m, beta, C.byref(a), C.byref(b)), "crd_steady_state")
    return a.value, b.value


def steady_state_as_printed(model, beta, decimals=8):
    """The stable state as the reference's Goldbeter programs read it from SolveGoldbeterODE.py's print (numpy: 8 decimals)."""
    a, b = C.c_double(), C.c_double()
    m = MODELS[model] if isinstance(model, str) else model
    check(lib().crd_steady_state_as_printed(m, beta, decimals, C.byref(a), C.byref(b)), "crd_steady_state_as_printed")
    return a.value, b.value


def stable_dt(params):
    return lib().crd_stable_dt(C.byref(params))


def load_ini(path, model, surface):
    cfg = RunConfig()
    err = C.create_string_buffer(512)
    m = MODELS[model] if isinstance(model, str) else model
    s = SURFACES[surface] if isinstance(surface, str) else surface
    rc = lib().crd_config_load_ini(str(path).encode(), m, s, C.byref(cfg), err, len(err))
    if rc != capi.OK:
        raise CrdError(rc, "crd_config_load_ini", err.value.decode())
    return cfg


def run_config(params, *, wave_length=0.1, wave_width=0.5, wave_inside=0, output_timestep=1, t_final=1.0,
               include_all_vars=0, ic_type=0, dt=0.0, dt_safety=0.8, n_gpus=1, stepper=capi.STEPPER_AUTO, adaptive=0, rtol=1e-5, atol=1e-10,
               steady_state_decimals=0):
    cfg = RunConfig()
    cfg.params = params
    cfg.wave_length, cfg.wave_width, cfg.wave_inside = wave_length, wave_width, wave_inside
    cfg.output_timestep, cfg.t_final = output_timestep, t_final
    cfg.include_all_vars, cfg.ic_type = include_all_vars, ic_type
    cfg.dt, cfg.dt_safety, cfg.n_gpus, cfg.stepper = dt, dt_safety, n_gpus, stepper
    cfg.adaptive, cfg.rtol, cfg.atol = adaptive, rtol, atol
    cfg.steady_state_decimals = steady_state_decimals
    return cfg


def initial_conditions(cfg, js=None, je=None, is_=None, ie=None):
    """Rows [js, je] (and columns [is_, ie]) of the reference's initial state as float64 (nyl, nxl, 2)."""
    g = grid_of(cfg.params)
    js = 0 if js is None else js
    je = g.ny - 1 if je is None else je
    is_ = 0 if is_ is None else is_
    ie = g.nx - 1 if ie is None else ie
    y = np.empty((je - js + 1, ie - is_ + 1, 2), dtype=np.float64)
    check(lib().crd_initial_conditions_block(C.byref(cfg), is_, ie, js, je, y.ctypes.data), "crd_initial_conditions_block")
    return y


class Writer:
    """Per-subdomain text files in the reference's format (/root/reference/src/FHNmodel_torus.cpp:376-410,438-455)."""

    def __init__(self, cfg, directory, slab=0, n_slabs=1, block=None):
        self._h = C.c_void_p()
        if block is None:
            check(lib().crd_writer_open(C.byref(cfg), str(directory).encode(), slab, n_slabs, C.byref(self._h)), "crd_writer_open")
        else:  # block = (c0, d0, c1, d1); the file number is the block's rank
            c0, d0, c1, d1 = block
            check(lib().crd_writer_open_block(C.byref(cfg), str(directory).encode(), c0 * d1 + c1, c0, d0, c1, d1, C.byref(self._h)), "crd_writer_open_block")

    def write_row(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        check(lib().crd_writer_write_row(self._h, y.ctypes.data), "crd_writer_write_row")

    def close(self):
        if self._h:
            h, self._h = self._h, None
            check(lib().crd_writer_close(h), "crd_writer_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _adaptive_options(options):
    opt = capi.AdaptiveOptions()
    check(lib().crd_adaptive_defaults(C.byref(opt)), "crd_adaptive_defaults")
    for k, v in options.items():
        if not hasattr(opt, k):
            raise TypeError("unknown adaptive option %r" % k)
        setattr(opt, k, v)
    return opt


def _adaptive_result(rc, st, where, handle):
    stats = {f: getattr(st, f) for f, _ in st._fields_}
    if rc != capi.OK:
        raise CrdError(rc, where, lib().crd_last_error(handle).decode() + " %r" % stats)
    return stats


class Slab:
    """One phi-slab of the grid resident on one GPU (crd_ctx)."""

    def __init__(self, params, slab=0, n_slabs=1, device=0, block=None):
        self._h = C.c_void_p()
        self.params = params
        if block is None:
            rc = lib().crd_create(C.byref(params), slab, n_slabs, device, C.byref(self._h))
        else:  # block = (c0, d0, c1, d1) of a 2-D decomposition
            c0, d0, c1, d1 = block
            slab, n_slabs = c0 * d1 + c1, d0 * d1
            rc = lib().crd_create_block(C.byref(params), c0, d0, c1, d1, device, C.byref(self._h))
        if rc != capi.OK:
            self._h = None
            raise CrdError(rc, "crd_create", lib().crd_last_error(None).decode())
        g = Grid()
        check(lib().crd_get_grid(self._h, C.byref(g)), "crd_get_grid", self._h)
        self.grid = g
        js, je = C.c_int64(), C.c_int64()
        check(lib().crd_get_slab(self._h, C.byref(js), C.byref(je)), "crd_get_slab", self._h)
        self.js, self.je = js.value, je.value
        blk = [C.c_int64() for _ in range(4)]
        check(lib().crd_get_block(self._h, *[C.byref(x) for x in blk]), "crd_get_block", self._h)
        self.is_, self.ie = blk[0].value, blk[1].value
        self.nx, self.nyl = self.ie - self.is_ + 1, je.value - js.value + 1
        self.slab, self.n_slabs = slab, n_slabs
        self.dtype = np.float64 if params.precision == PRECISION_F64 else np.float32

    # -- lifecycle ---------------------------------------------------------------------------------------------
    def close(self):
        if self._h:
            lib().crd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def _check(self, rc, where):
        check(rc, where, self._h)

    # -- state -------------------------------------------------------------------------------------------------
    def upload(self, y):
        y = np.ascontiguousarray(y)
        assert y.shape == (self.nyl, self.nx, 2), (y.shape, (self.nyl, self.nx, 2))
        if y.dtype == np.float64:
            self._check(lib().crd_state_upload(self._h, y.ctypes.data, 1), "crd_state_upload")
        elif y.dtype == np.float32 and self.dtype == np.float32:
            self._check(lib().crd_state_upload(self._h, y.ctypes.data, 0), "crd_state_upload")
        else:
            raise TypeError("state must be float64, or float32 for an f32 context")

    def download(self, dtype=np.float64):
        y = np.empty((self.nyl, self.nx, 2), dtype=dtype)
        self._check(lib().crd_state_download(self._h, y.ctypes.data, 1 if y.dtype == np.float64 else 0), "crd_state_download")
        return y

    def download_rows(self, var, row_begin, row_count):
        """Rows [row_begin, row_begin + row_count) of one field of the resident state, ghost rows included (device precision)."""
        rows = np.empty((row_count, self.nx), dtype=self.dtype)
        self._check(lib().crd_state_download_rows(self._h, var, row_begin, row_count, rows.ctypes.data), "crd_state_download_rows")
        return rows

    # -- the RHS callback --------------------------------------------------------------------------------------
    def f(self, t, y, out=None):
        """ydot = f(t, y) for this slab's AoS vector (host arrays in the device precision).  With y and `out` both over
        pinned memory (PinnedArray) the call streams the slab band by band over the host link."""
        y = np.ascontiguousarray(y, dtype=self.dtype)
        assert y.shape == (self.nyl, self.nx, 2)
        ydot = np.empty_like(y) if out is None else out
        assert ydot.shape == y.shape and ydot.dtype == y.dtype and ydot.flags["C_CONTIGUOUS"]
        self._check(lib().crd_rhs_host(self._h, t, y.ctypes.data, ydot.ctypes.data), "crd_rhs_host")
        return ydot

    # -- time stepping -----------------------------------------------------------------------------------------
    def set_stepper(self, stepper):
        s = STEPPERS[stepper] if isinstance(stepper, str) else stepper
        self._check(lib().crd_set_stepper(self._h, s), "crd_set_stepper")

    def step_rk4(self, t0, dt, nsteps, sync=True):
        self._check(lib().crd_step_rk4(self._h, t0, dt, nsteps), "crd_step_rk4")
        if sync:
            self.synchronize()

    def step_rk4_timed(self, t0, dt, nsteps):
        ms, kms, lps = C.c_double(), C.c_double(), C.c_int()
        self._check(lib().crd_step_rk4_timed(self._h, t0, dt, nsteps, C.byref(ms), C.byref(kms), C.byref(lps)), "crd_step_rk4_timed")
        return ms.value, kms.value, lps.value

    def integrate_adaptive(self, t0, tout, **options):
        """Error-controlled integration from t0 to tout; options override crd_adaptive_defaults (rtol, atol, h0, method, ...): by
        default the reference's integrator (ARKode's Zonneveld 5(3)4 pair + PID controller, ARK_NORMAL output); method=0 with
        dense_output=0/1 is the RK4(3) pair of earlier rounds.  Returns stats."""
        opt, st = _adaptive_options(options), capi.AdaptiveStats()
        rc = lib().crd_integrate_adaptive(self._h, t0, tout, C.byref(opt), C.byref(st))
        return _adaptive_result(rc, st, "crd_integrate_adaptive", self._h)

    def synchronize(self):
        self._check(lib().crd_synchronize(self._h), "crd_synchronize")

    def set_diagnostics(self, on):
        """Per-exchange event pairs in the next step_rk4_timed calls (RCCL contexts): see step_timing()."""
        self._check(lib().crd_set_diagnostics(self._h, 1 if on else 0), "crd_set_diagnostics")

    def set_halo_slack(self, sweeps):
        """1 or 2 sweeps of owned-only rows between an exchange and the wait for its halo (crd_set_halo_slack)."""
        self._check(lib().crd_set_halo_slack(self._h, sweeps), "crd_set_halo_slack")

    def set_exchange_period(self, steps):
        """Fused steps between two deep-halo exchanges (crd_set_exchange_period): 3 .. 16, the same on every slab / rank of a run."""
        self._check(lib().crd_set_exchange_period(self._h, steps), "crd_set_exchange_period")

    def exchange_period(self):
        return lib().crd_get_exchange_period(self._h)

    def step_timing(self):
        """What the last step_rk4_timed call measured (crd_step_timing) as a dict."""
        tm = capi.StepTiming()
        self._check(lib().crd_get_step_timing(self._h, C.byref(tm)), "crd_get_step_timing")
        return {f: getattr(tm, f) for f, _ in tm._fields_ if f != "reserved"}

    def dominant_kernel_rows(self):
        v = C.c_int64()
        self._check(lib().crd_dominant_kernel_rows(self._h, C.byref(v)), "crd_dominant_kernel_rows")
        return v.value

    def dominant_kernel(self):
        return lib().crd_dominant_kernel_name(self._h).decode()

    def set_autotune(self, on):
        """0 / False: never measure a launch plan; 1 / True: measure on the first full-size launch; 2: ... and print the timings."""
        self._check(lib().crd_set_autotune(self._h, int(on)), "crd_set_autotune")

    def set_launch_plan(self, chunk_mode, xcd_mapping, columns_per_lane, nontemporal_stores=0, steps_per_launch=1):
        """Pin the launch plan of the fixed-step kernel (crd_set_launch_plan) instead of having it measured."""
        self._check(lib().crd_set_launch_plan(self._h, chunk_mode, xcd_mapping, columns_per_lane, nontemporal_stores, steps_per_launch), "crd_set_launch_plan")

    def plan_launches(self):
        """Measure the fused step kernel's launch plan now (state not advanced) rather than inside the first step_rk4."""
        self._check(lib().crd_plan_launches(self._h), "crd_plan_launches")

    def launch_plan(self):
        """The fused step kernel's launch plan as a dict (crd_launch_plan)."""
        lp = capi.LaunchPlan()
        self._check(lib().crd_get_launch_plan(self._h, C.byref(lp)), "crd_get_launch_plan")
        return {f: getattr(lp, f) for f, _ in lp._fields_ if f != "reserved"}

    def launch_geometry(self):
        """The full-height launch under the current plan + what the build's kernel table says about its kernel (crd_launch_geometry)."""
        g = capi.LaunchGeometry()
        self._check(lib().crd_get_launch_geometry(self._h, C.byref(g)), "crd_get_launch_geometry")
        return {f: getattr(g, f) for f, _ in g._fields_ if f != "reserved"}

    def max_abs(self):
        v = C.c_double()
        self._check(lib().crd_state_max_abs(self._h, C.byref(v)), "crd_state_max_abs")
        return v.value

    # -- RCCL wiring -------------------------------------------------------------------------------------------
    def init_rccl(self, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(lib().crd_comm_init_rccl(self._h, buf), "crd_comm_init_rccl")

    def comm_info(self):
        """(transport, ranks, rank): transport is 'self' / 'local' / 'rccl'; for RCCL the last two come from the communicator."""
        halo, ranks, rank = C.c_int(), C.c_int(), C.c_int()
        self._check(lib().crd_comm_info(self._h, C.byref(halo), C.byref(ranks), C.byref(rank)), "crd_comm_info")
        return {0: "self", 1: "local", 2: "rccl"}.get(halo.value, "unwired"), ranks.value, rank.value

    def halo_exchange(self, depth=32):
        """One exchange of `depth` ghost rows of both fields with the ring neighbours, outside any step; waits for it."""
        self._check(lib().crd_halo_exchange(self._h, depth), "crd_halo_exchange")


class PinnedArray:
    """A numpy array over page-locked host memory from crd_host_alloc (freed with the object)."""

    def __init__(self, shape, dtype=np.float64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().crd_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("crd_host_alloc(%d) failed" % self.nbytes)
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p), dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().crd_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def launch_plan_candidates():
    """The plans the launch-plan measurement chooses among, as (chunk_mode, xcd_mapping, columns_per_lane, nontemporal_stores)."""
    out, k = [], 0
    while True:
        lp = capi.LaunchPlan()
        if lib().crd_launch_plan_candidate(k, C.byref(lp)) != 0:
            return out
        out.append((lp.one_round, lp.xcd_mapping, lp.columns_per_lane, lp.nontemporal_stores, lp.steps_per_launch))
        k += 1


def plan_key(model, precision, plan):
    """Key of a launch plan in profiles/pmc_traffic.json / profiles/plan_stats.json: plan = (chunk_mode, mapping, columns, nt[, steps per
    launch]) or the dict Slab.launch_plan() returns."""
    if isinstance(plan, dict):
        plan = (plan["one_round"], plan["xcd_mapping"], plan["columns_per_lane"], plan["nontemporal_stores"], plan.get("steps_per_launch", 1))
    key = "fused/%s/%s/chunk%d/map%d/cols%d/%s" % (model, precision, plan[0], plan[1], plan[2], "nt" if plan[3] else "plain")
    return key + ("/steps%d" % plan[4] if len(plan) > 4 and plan[4] >= 2 else "")


def kernel_table_digest():
    """crd_kernel_table_digest: 16 hex digits over the step kernels of the loaded library ("" for a build without the table)."""
    return lib().crd_kernel_table_digest().decode()


def kernel_digest(geometry):
    """16 hex digits over what the build's kernel table says about ONE kernel -- the one `geometry` (Slab.launch_geometry()) describes:
    registers, LDS, scratch, occupancy and the instruction mix of its steady-state loop.  profiles/pmc_traffic.json and plan_stats.json
    stamp every entry with the digest of the kernel it was measured on; bench.py quotes an entry only while the loaded library's kernel
    still has that digest (a changed kernel makes its own entries stale, not the other kernels')."""
    import hashlib

    fields = ("vgprs", "sgprs", "lds_bytes", "scratch_bytes", "wavefronts_per_simd", "loop_valu", "loop_salu", "loop_vmem", "loop_lds", "loop_instructions", "exec_skipped_vmem")
    if not geometry or not geometry.get("vgprs"):
        return ""
    return hashlib.sha256(",".join("%s=%d" % (f, int(geometry[f])) for f in fields).encode()).hexdigest()[:16]


def kernel_digest_of_table_row(row):
    """The same digest from a row of csrc/build/kernel_table.json (tools/kernel_regs.py --json): what a build's kernel would report
    through crd_get_launch_geometry, without a device (tests/test_profiles.py)."""
    lp = row["loop"]
    return kernel_digest({"vgprs": row["vgprs"], "sgprs": row["sgprs"], "lds_bytes": row["lds_bytes"], "scratch_bytes": row["scratch_bytes"], "wavefronts_per_simd": row["wavefronts_per_simd"],
                          "loop_valu": lp["valu"], "loop_salu": lp["salu"], "loop_vmem": lp["vmem"], "loop_lds": lp["lds"], "loop_instructions": lp["total"],
                          "exec_skipped_vmem": row.get("exec_skipped_vmem", 0)})


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().crd_comm_unique_id(buf), "crd_comm_unique_id")
    return buf.raw


class LocalGroup:
    """All slabs of one run inside this process (one host thread drives every GPU, or several slabs on one GPU).  blocks = (d0, d1):
    the reference's 2-D decomposition instead of phi-slabs, d0 theta-blocks x d1 phi-blocks, in rank order c0 d1 + c1."""

    def __init__(self, params, n_slabs, devices=None, blocks=None):
        if blocks is not None:
            d0, d1 = blocks
            n_slabs = d0 * d1
        devices = devices or [0] * n_slabs
        if blocks is None:
            self.slabs = [Slab(params, k, n_slabs, devices[k]) for k in range(n_slabs)]
        else:
            self.slabs = [Slab(params, device=devices[c0 * d1 + c1], block=(c0, d0, c1, d1)) for c0 in range(d0) for c1 in range(d1)]
        self._arr = (C.c_void_p * n_slabs)(*[s.handle for s in self.slabs])
        check(lib().crd_comm_attach_local(self._arr, n_slabs), "crd_comm_attach_local", self.slabs[0].handle)
        self.grid = self.slabs[0].grid

    def upload(self, y):
        for s in self.slabs:
            s.upload(y[s.js:s.je + 1, s.is_:s.ie + 1])

    def _assemble(self, parts, dtype):
        out = np.empty((self.grid.ny, self.grid.nx, 2), dtype=dtype)
        for s, q in zip(self.slabs, parts):
            out[s.js:s.je + 1, s.is_:s.ie + 1] = q
        return out

    def download(self, dtype=np.float64):
        return self._assemble([s.download(dtype) for s in self.slabs], dtype)

    def f(self, t, y):
        """ydot = f(t, y) on the whole grid, evaluated block by block with halos exchanged between the blocks' vectors."""
        n = len(self.slabs)
        parts = [np.ascontiguousarray(y[s.js:s.je + 1, s.is_:s.ie + 1], dtype=s.dtype) for s in self.slabs]
        outs = [np.empty_like(q) for q in parts]
        ins = (C.c_void_p * n)(*[q.ctypes.data for q in parts])
        out = (C.c_void_p * n)(*[q.ctypes.data for q in outs])
        check(lib().crd_group_rhs_host(self._arr, n, t, ins, out), "crd_group_rhs_host", self.slabs[0].handle)
        return self._assemble(outs, outs[0].dtype)

    def step_rk4(self, t0, dt, nsteps):
        check(lib().crd_group_step_rk4(self._arr, len(self.slabs), t0, dt, nsteps), "crd_group_step_rk4", self.slabs[0].handle)
        for s in self.slabs:
            s.synchronize()

    def step_rk4_timed(self, t0, dt, nsteps):
        """crd_group_step_rk4_timed: the group call with event pairs on every slab; returns each slab's step_timing()."""
        check(lib().crd_group_step_rk4_timed(self._arr, len(self.slabs), t0, dt, nsteps), "crd_group_step_rk4_timed", self.slabs[0].handle)
        return [s.step_timing() for s in self.slabs]

    def set_stepper(self, stepper):
        for s in self.slabs:
            s.set_stepper(stepper)

    def set_halo_slack(self, sweeps):
        for s in self.slabs:
            s.set_halo_slack(sweeps)

    def set_exchange_period(self, steps):
        for s in self.slabs:
            s.set_exchange_period(steps)

    def set_threads(self, threads):
        """Host threads issuing the group's work in step_rk4 (crd_group_set_threads): 0 = one per device."""
        check(lib().crd_group_set_threads(self._arr, len(self.slabs), threads), "crd_group_set_threads", self.slabs[0].handle)

    def integrate_adaptive(self, t0, tout, **options):
        opt, st = _adaptive_options(options), capi.AdaptiveStats()
        rc = lib().crd_group_integrate_adaptive(self._arr, len(self.slabs), t0, tout, C.byref(opt), C.byref(st))
        stats = _adaptive_result(rc, st, "crd_group_integrate_adaptive", self.slabs[0].handle)
        for s in self.slabs:
            s.synchronize()
        return stats

    def close(self):
        for s in self.slabs:
            s.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
