#!/usr/bin/env python3
"""bench.py -- grid-point-steps/s of the HIP RK4 path on the BASELINE.json workload (FHN torus 8192^2, fp64).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one classical RK4 step of the whole grid on synthetic initial conditions made by the reference's own rule; the state
is resident in HBM before the timed region; the grid is fixed as N grows (strong scaling); rank r owns phi-slab r of N; rank 0
prints ONE JSON line.  Order of a run: set-up and halo self-check, launch-plan measurement (crd_plan_launches, or --launch-plan),
rehearsals (N > 1), --preheat-ms of untimed stepping, W warm-up steps, K timed steps between two fences (MAX over ranks).

How N > 1 runs -- one state machine, whoever starts it:

    START --(WORLD_SIZE unset, N > 1, transport != local)--> SELF-LAUNCH: this process touches no GPU, starts N rank processes of
          itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1, a free port) and watches ALL of them
    RANK  (started by us or by torch.distributed.run): roll call over gloo -> RING BRING-UP in a helper thread the rank stops
          waiting for after --ring-timeout-s (ncclCommInitRank, communicator self-report, 32-row halo self-check) -> the ranks
          tell each other how it went (AGREED OUTCOME) -> ok: rehearsals, timed region, rank 0 prints the line
    ring failed anywhere, --transport auto -> LOCAL LEG: ONE fresh child process steps all N slabs as a LOCAL group (peer copies,
          one issuing host thread per GPU).  Under our launcher the ranks leave with status 4 and the launcher starts that child;
          under an external launcher ranks 1.. leave with status 0 and rank 0 starts it and passes its line on.  A process that has
          touched the GPU is never replaced.  config.launcher says which way the line came; --transport rccl / local pin a leg.
    The legs share one time budget (--launch-timeout-s in all): the LOCAL leg gets what the ring's leg left.

`roofline` prices the dominant kernel by ITS OWN compulsory bytes (the one-launch step: read + write of the state once per
launch, 4 reals per point), measured live with HIP events on the library's stream; `issue_frac` is the same launch on the
vector-issue roof, from the instruction count of the kernel's loop in the build's own assembly (crd_get_launch_geometry);
`bound` names the roof the launch sits closer to.  SURVEY 8(d)'s 32-reals-per-step scheme is measured in `staged`.
PyTorch is plumbing only: a GLOO group on CPU tensors; each rank owns ONE RCCL communicator, libcrd's.  The oracle is used only
by the `cpu_baseline` leg."""
import argparse
import json
import os
import subprocess
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from crdmodel_amd.benchkit import ControlPlane, committed_json, halo_selfcheck, measured_traffic, selfcheck_ghost_rows, selfcheck_pattern, stale_reason, usable_cores  # noqa: E402,F401 (no GPU, no libcrd yet)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
REALS_PER_POINT_STEP = 32  # 6 (stage 1) + 10 + 10 (stages 2, 3) + 6 (stage 4) reals per grid-point-step, staged scheme
REALS_PER_POINT_STAGE23 = 10
LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GLOO_SOCKET_IFNAME",
                 "CRD_BENCH_SELF_LAUNCHED")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=8192, help="theta and phi mesh (BASELINE: 8192)")
    ap.add_argument("--model", default="fhn", choices=["fhn", "goldbeter"])
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--stepper", default="auto", choices=["auto", "staged", "fused"])
    ap.add_argument("--dt", type=float, default=0.0, help="RK4 step; default 0.8 x the stability limit of the grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--t-boundary", type=float, default=0.0, help="tBoundary: > 0 keeps the absorbing rows on for the whole run (SURVEY 8d's extra run)")
    ap.add_argument("--cpu-rows", type=int, default=4096, help="phi rows of the grid the CPU baseline integrates")
    ap.add_argument("--cpu-steps", type=int, default=72, help="RK4 steps of the CPU baseline (about 12 s on 16 cores)")
    ap.add_argument("--force-rccl", action="store_true", help="world size 1 only: route the halos through an RCCL ring to self (rehearses the N>1 code path)")
    ap.add_argument("--halo-slack-threshold-ms", type=float, default=0.03, help="ring runs: exposed wait per exchange (max over ranks) above which the halo gets a third sweep of cover")
    ap.add_argument("--preheat-ms", type=float, default=300.0, help="untimed stepping in front of the warm-up, so that a short timed region runs at the clocks a long run settles at")
    ap.add_argument("--launch-plan", default="", metavar="MODE,MAPPING,COLUMNS[,NT[,STEPS]]", help="pin the step kernel's launch plan (crd_set_launch_plan) instead of measuring it")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "local"], help="N > 1: rccl = one rank process per GPU; local = one process, LOCAL group; auto = rccl, then local")
    ap.add_argument("--ring-timeout-s", type=float, default=150.0, help="a rank stops waiting for the RCCL ring's bring-up after this long (0 = wait for ever)")
    ap.add_argument("--launch-timeout-s", type=float, default=420.0, help="time budget of ALL legs of an N > 1 run together (a healthy 8-rank run takes about a minute)")
    ap.add_argument("--exchange-period", type=int, default=0, help="ring runs: fused steps per deep-halo exchange; 0 = rehearse 8, 10 and 16, keep the fastest")
    ap.add_argument("--devices", default="", help="--transport local: device ordinal of each slab, e.g. 0,0 to rehearse two slabs on one GPU (default 0..N-1)")
    ap.add_argument("--issuing-threads", type=int, default=0, help="--transport local: host threads issuing the group's work (0 = one per device)")
    ap.add_argument("--crd-module", default="crdmodel_amd", help=argparse.SUPPRESS)  # tests: a stand-in for the device API (tests/standin_crd.py)
    ap.add_argument("--repeats", type=int, default=4, help="the K timed steps again, this many times, AFTER the timed region (spread of the figure; not part of `value`)")
    ap.add_argument("--one-step-steps", type=int, default=40, help="steps of the one-step-per-launch kernel timed beside a two-step run (0 = skip)")
    ap.add_argument("--staged-steps", type=int, default=40, help="steps of the staged stepper timed beside a fused run for the `staged` sub-record (0 = skip)")
    ap.add_argument("--preflight", action="store_true", help="N ranks: roll call, RCCL ring bring-up, communicator self-report and the 32-row halo self-check ONLY; one JSON "
                                                             "line ({\"preflight\": true, \"ok\": ...}) within a minute -- why a multi-GPU run would fail, without running it")
    return ap.parse_args(argv)


def cpu_baseline(args, beta, dt):
    """The oracle (a port of the reference's f() + classical RK4) timed on this box's host cores, on a bounded band."""
    from oracle import crd_oracle as co

    threads, rows = usable_cores(), min(args.cpu_rows, args.size)
    model = co.FHN if args.model == "fhn" else co.GOLDBETER

    def band(ny, steps, nthreads):
        op = co.make_problem(model, co.TORUS, args.size, 80.0, 20.0, 0.12, beta, ny=ny)
        op.dy = (2.0 * 3.1415926535897932) / (1.0 * args.size - 1.0)  # the band keeps the full grid's phi spacing
        y0 = co.initial_conditions(op, 0.1, 0.5, 0, 0)
        if nthreads > 1:
            co.rk4(op, y0, 0.0, dt, 1, nthreads=nthreads)  # warm-up (page faults, OpenMP pool)
        t0 = time.perf_counter()
        co.rk4(op, y0, 0.0, dt, steps, nthreads=nthreads)
        return time.perf_counter() - t0

    el = band(rows, args.cpu_steps, threads)
    rows1, steps1 = max(16, rows // 8), max(1, args.cpu_steps // 18)  # the same code on ONE core (SURVEY 8d asks for both figures)
    el1 = band(rows1, steps1, 1)
    return {"value": args.size * rows * args.cpu_steps / el, "unit": "grid-point-steps/s", "cores": threads, "kind": "port",
            "sample": "%dx%d band of the %dx%d grid, %d RK4 steps, faithful per-point sin/cos RHS (oracle/crd_oracle.c), OpenMP over rows, %.1f s"
                      % (args.size, rows, args.size, args.size, args.cpu_steps, el),
            "value_1core": args.size * rows1 * steps1 / el1, "sample_1core": "%dx%d band, %d RK4 steps, one thread, %.1f s" % (args.size, rows1, steps1, el1)}


# ---- the launcher side of the state machine (module docstring) ---------------------------------------------------------------------
def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def last_json_line(text):
    for ln in reversed((text or "").strip().splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def without_transport(argv):
    out, skip = [], False
    for a in argv:
        if skip or a == "--transport" or a.startswith("--transport="):
            skip = (a == "--transport")
            continue
        out.append(a)
    return out


def end_processes(procs):
    """End exactly the processes we started that are still there (by PID, never by pattern)."""
    for pr in procs:
        if pr.poll() is None:
            pr.terminate()
    for pr in procs:
        try:
            pr.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()


def ring_leg(args, argv, deadline):
    """SELF-LAUNCH: N rank processes of this script, one per GPU, watched together: the first rank that exits non-zero ends the leg at
    once (its siblings would sit in a gloo collective until their time-outs).  Returns (rank 0's line or None, reason)."""
    n = args.gpus
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", CRD_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--transport", "rccl"]
    out0 = open(os.path.join("/tmp", "crd_bench_rank0_%d.out" % os.getpid()), "w+")
    procs = [subprocess.Popen(cmd, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=out0 if r == 0 else subprocess.DEVNULL) for r in range(n)]
    why = ""
    while any(pr.poll() is None for pr in procs):
        if any(pr.returncode not in (None, 0) for pr in procs):
            break  # somebody failed: no point in waiting for the others' time-outs
        if time.monotonic() > deadline:
            why = "rank processes did not finish within the time budget"
            break
        time.sleep(0.25)
    end_processes(procs)
    codes = [pr.returncode for pr in procs]
    out0.seek(0)
    line = last_json_line(out0.read())
    out0.close()
    os.unlink(out0.name)
    why = why or ("rank exit codes %s" % codes if any(codes) else ("rank 0 printed no JSON line" if line is None else ""))
    return (None if why else line), why


def local_leg(argv, deadline):
    """LOCAL LEG: one fresh child process for all GPUs (no launcher variables in its environment).  Returns (line or None, status, reason)."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + without_transport(argv) + ["--transport", "local"], env=env, stdout=subprocess.PIPE, text=True,
                           timeout=max(20.0, deadline - time.monotonic()))
    except subprocess.TimeoutExpired:
        return None, 1, "local: did not finish within the time budget"
    line = last_json_line(r.stdout)
    return line, (r.returncode if line is not None else (r.returncode or 1)), ("" if line is not None else "local: exit code %d, no JSON line" % r.returncode)


def launch(args, argv):
    """START without a launcher, N > 1: be the launcher.  Returns the exit status."""
    argv = without_transport(sys.argv[1:] if argv is None else list(argv))
    deadline = time.monotonic() + args.launch_timeout_s
    tried, reasons, line, rc = [], [], None, 1
    if args.preflight:
        # the ring's leg alone, a minute at most, and a line either way: rank 0's on success, ours with the reason otherwise
        t0 = time.monotonic()
        line, why = ring_leg(args, argv, t0 + min(60.0, args.launch_timeout_s))
        if line is None:
            line = {"preflight": True, "ok": False, "n_gpus": args.gpus, "reasons": [why], "elapsed_s": round(time.monotonic() - t0, 2)}
        line["launcher"] = "self-launched: %d rank process(es) started by bench.py itself" % args.gpus
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()
        return 0 if line.get("ok") else 1
    if args.transport in ("auto", "rccl"):
        tried.append("rccl")
        line, why = ring_leg(args, argv, deadline - (60.0 if args.transport == "auto" else 0.0))  # (auto: the LOCAL leg keeps a minute of the budget at least)
        rc = 0 if line is not None else 1
        if line is None:
            reasons.append("rccl: " + why)
            sys.stderr.write("bench.py: the rccl leg failed (%s)%s\n" % (why, "; trying --transport local" if args.transport == "auto" else ""))
    if line is None and args.transport in ("auto", "local"):
        tried.append("local")
        line, rc, why = local_leg(argv, deadline)
        if why:
            reasons.append(why)
    if line is None:
        raise SystemExit("bench.py --gpus %d: no leg produced a result (%s)" % (args.gpus, "; ".join(reasons)))
    ranks = args.gpus if line["config"].get("halo", {}).get("transport") == "rccl" else 1
    line.setdefault("config", {})["launcher"] = {"mode": "self-launched: %d rank process(es) started by bench.py itself" % ranks, "transports_tried": tried, "fallback_reasons": reasons}
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()
    return rc


def ring_failed(args, ctl, slab, rank, world, failures, stuck, emit, t_start):
    """AGREED OUTCOME = failed (every rank knows: the statuses went over the control plane).  Our own launcher, or --transport rccl: every
    rank leaves with status 4.  An external launcher with --transport auto: ranks 1.. leave with status 0, rank 0 runs the LOCAL LEG in a
    child and passes its line on.  A rank whose bring-up thread is stuck inside RCCL cannot tear anything down: it leaves through os._exit."""
    why = "; ".join(failures)
    if args.preflight:
        if rank == 0:
            emit(json.dumps({"preflight": True, "ok": False, "n_gpus": world, "reasons": failures, "elapsed_s": round(time.monotonic() - t_start, 2)}))
        if not stuck:
            slab.close()
        ctl.close()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(4) if stuck else sys.exit(4)
    relay = args.transport == "auto" and not os.environ.get("CRD_BENCH_SELF_LAUNCHED")
    if not stuck:
        slab.close()
    ctl.close()
    code = 0 if relay else 4
    if relay and rank == 0:
        sys.stderr.write("bench.py: the rccl leg failed (%s); rank 0 runs the LOCAL leg in a child process\n" % why)
        line, code, lwhy = local_leg(sys.argv[1:], t_start + args.launch_timeout_s)
        if line is not None:
            line.setdefault("config", {})["launcher"] = {"mode": "%d rank processes started by an external launcher; the ring failed, rank 0 ran the LOCAL leg in a child process" % world,
                                                          "transports_tried": ["rccl", "local"], "fallback_reasons": ["rccl: " + why]}
            emit(json.dumps(line))
        else:
            sys.stderr.write("bench.py: no leg produced a result (rccl: %s; %s)\n" % (why, lwhy))
    elif rank == 0:
        sys.stderr.write("bench.py: set-up of the ring failed: %s\n" % why)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code) if stuck else sys.exit(code)


def main(argv=None):
    args = parse(argv)
    t_start = time.monotonic()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver only supports dmabuf IPC (RCCL peer buffers)
    os.environ.setdefault("NCCL_DEBUG", "WARN")  # should the ring fail at first contact, RCCL says why on stderr
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.transport != "local":
        sys.exit(launch(args, argv))  # (before torch is imported or HIP is touched: the rank processes are fresh children)
    # ONE JSON line on stdout: RCCL prints a banner to fd 1 when a communicator is created, so everything else goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import importlib

    standin = args.crd_module != "crdmodel_amd"
    if not standin:
        import torch  # first: its bundled HIP runtime is the one this process uses

        lib = os.path.join(ROOT, "crdmodel_amd", "libcrd.so")
        if not os.path.exists(lib) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
            from crdmodel_amd.build import build  # a checkout without the (untracked) built library: build it in-tree first

            build()
        for _ in range(600):  # the other local ranks wait for rank 0's build
            if os.path.exists(lib):
                break
            time.sleep(0.5)
    crd = importlib.import_module(args.crd_module)

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not standin and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libcrd has no CPU fallback")
    if args.transport == "local" and args.gpus > 1:
        if world != 1:
            raise SystemExit("--transport local is ONE process for all GPUs: start it without a launcher")
        return run_local(args, crd, 0 if standin else torch.cuda.device_count(), emit)
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if standin:
        return run(args, crd, world, rank, local_rank, lambda: None, emit, t_start)
    local_rank %= max(1, torch.cuda.device_count())  # a launcher that narrows the visible devices per rank leaves one
    torch.cuda.set_device(local_rank)
    run(args, crd, world, rank, local_rank, torch.cuda.synchronize, emit, t_start)


def problem(args, crd):
    beta = 1.25 if args.model == "fhn" else 0.4
    params = crd.make_params(args.model, "torus", args.size, 80.0, 20.0, 0.12, beta, ny=args.size, t_boundary=args.t_boundary, precision=args.precision)
    dt = args.dt if args.dt > 0 else 0.8 * crd.stable_dt(params)
    return beta, params, dt, crd.run_config(params, wave_length=0.1, wave_width=0.5, wave_inside=0)


def issue_model(geo, kernel_ms):
    """The timed launch on the vector-issue roof: (vector instructions of one trip of the kernel's steady-state loop, counted in the build's
    own assembly) x trips of all wavefronts x 4 cycles per wavefront instruction, over the device's SIMDs x clock x the launch's time."""
    if not geo or not geo.get("loop_valu") or kernel_ms <= 0:
        return None
    trips = geo["wavefront_iterations_effective"] / geo["iterations_per_trip"]  # (an item's filling iterations at the stages they actually run)
    floor_ms = geo["loop_valu"] * trips * 4.0 / (geo["simds"] * geo["clock_khz"] * 1e3) * 1e3
    return {"issue_frac": floor_ms / kernel_ms, "issue_floor_ms": floor_ms, "valu_instructions_per_trip": geo["loop_valu"], "iterations_per_trip": geo["iterations_per_trip"],
            "wavefront_iterations": geo["wavefront_iterations"], "wavefront_iterations_effective": geo["wavefront_iterations_effective"], "cycles_per_wavefront_instruction": 4, "simds": geo["simds"], "clock_mhz": geo["clock_khz"] / 1e3,
            "vgprs": geo["vgprs"], "wavefronts_per_simd": geo["wavefronts_per_simd"], "lanes_valid": "%d of %d" % (geo["lanes_valid"], geo["lanes"]),
            "chunk_rows": geo["chunk_rows"], "fill_iterations_per_item": geo["fill_iterations"],
            "source": "crd_get_launch_geometry: loop instruction mix from the assembly of this build's kernel (tools/kernel_regs.py), launch geometry of this run's plan, "
                      "hipDeviceProp_t clock; an upper bound on the clock makes it a LOWER bound on the fraction"}


def build_line(args, crd, world, dt, beta, elapsed, kernel, kernel_ms, launches, pts_launch, device_ms_per_step, plan, comm, preheat, staged, per_rank, kernel_ms_source,
               per_launch=None, geometry=None):
    """The JSON line (a dict) of a finished run.  elapsed: wall time of the K timed steps (MAX over ranks); kernel_ms: average duration
    of one launch of the dominant kernel on rank 0, covering pts_launch grid points and advancing them per_launch steps."""
    n, real, fused = args.size, 8 if args.precision == "f64" else 4, launches == 1
    value, ms_per_step = n * n * args.steps / elapsed, elapsed * 1e3 / args.steps

    def rate(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

    if args.launch_plan:
        plan["pinned"] = True  # (--launch-plan: nothing was measured in this run)
    key = crd.plan_key(args.model, args.precision, plan) if fused else "stage23/%s/%s" % (args.model, args.precision)
    per_launch = (per_launch or int(plan.get("steps_per_launch", 1))) if fused else 1  # steps the TIMED launches advanced (crd_step_timing)
    # the committed profiler records are quoted only for the kernel they were measured on: every entry carries the digest of its kernel
    # (registers and loop instruction mix as the assembler printed them), compared with the loaded library's (crd_get_launch_geometry)
    digest = crd.kernel_digest(geometry) if (fused and geometry is not None and hasattr(crd, "kernel_digest")) else None
    traffic, traffic_source = measured_traffic(key, pts_launch, digest)
    if fused:
        reals_launch = 4
        bytes_model = ("fused step kernel (all RK4 stages of a step in one launch): compulsory bytes per launch = read + write of both fields once = 4 reals per "
                       "point (%d B) x the points the launch covers%s" % (4 * real, "; this launch advances its points by %s steps (%.1f B per grid-point-step)"
                                                                           % ({2: "TWO", 3: "THREE"}.get(per_launch, str(per_launch)), 4.0 * real / per_launch) if per_launch >= 2 else ""))
    else:
        reals_launch = REALS_PER_POINT_STAGE23
        bytes_model = ("stage-2/3 kernel of the staged stepper: reads y_stage, y_n, acc and writes acc, y_next = 10 reals per point (%d B), SURVEY 8(d); a whole "
                       "step is 6 + 10 + 10 + 6 = 32 reals per point" % (REALS_PER_POINT_STAGE23 * real))
    alg_bytes = reals_launch * real * pts_launch
    achieved = rate(alg_bytes, kernel_ms)
    # the same accounting on the clock `value` uses: the compulsory bytes of ONE STEP of the whole grid over the wall time per step, per GPU
    step_bytes = (4.0 / per_launch if fused else REALS_PER_POINT_STEP) * real * n * n
    achieved_wall = rate(step_bytes, ms_per_step) / world
    streams = committed_json("hbm_streams.json")
    issue = issue_model(geometry, kernel_ms) if fused else None
    # which roof the launch sits closer to: its compulsory traffic against what this device streams at all (committed measurement; the
    # spec peak is never reached), its vector issue against the SIMDs' issue rate
    stream_gbs = 0.5 * (streams.get("read_only_gbs", 0) + streams.get("write_only_gbs", streams.get("read_only_gbs", 0))) if streams else 0.0
    hbm_of_streaming = achieved / stream_gbs if stream_gbs else achieved / HBM_PEAK_GBS
    bound = "valu-issue" if issue and issue["issue_frac"] > 1.05 * hbm_of_streaming else "hbm"  # (within 5 % of each other: the probe builds say memory, see bound_note)
    roofline = {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_wall": achieved_wall / HBM_PEAK_GBS,
                "achieved_wall": achieved_wall, "traffic": traffic, "traffic_source": traffic_source, "bytes_model": bytes_model, "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kernel_ms, "kernel_ms_source": kernel_ms_source, "launches_per_step": launches, "device_ms_per_step": device_ms_per_step, "plan_key": key,
                "frac_of_device_streaming": hbm_of_streaming if stream_gbs else None, "kernel_digest": digest,
                "kernel_table_digest": crd.kernel_table_digest() if hasattr(crd, "kernel_table_digest") else None}
    if issue:
        roofline.update({"issue_frac": issue["issue_frac"], "issue": issue})
        roofline["bound_note"] = ("`bound` = the roof the launch sits closer to: frac_of_device_streaming (compulsory bytes over what this device streams, "
                                  "profiles/hbm_streams.json) against issue_frac (vector instructions x 4 cycles over the SIMDs' time).  Probe builds "
                                  "(profiles/r06/three_step_ab.txt): the two-step launch as a copy takes 0.96 - 1.0 of the full launch's time and its arithmetic alone 0.75; "
                                  "the three-step launch as a copy 0.73, its arithmetic alone 0.92")
        busy = committed_json("valu_busy.json").get("%s/%s/cols%d/steps%d" % (args.model, args.precision, int(plan.get("columns_per_lane", 1)), per_launch))
        if busy:  # committed SQ counters of the same instantiation, for comparison with issue_frac
            roofline["issue"]["sq_counters"] = busy
    stats = committed_json("plan_stats.json").get(key)
    if stats:  # the committed `rocprofv3 --kernel-trace --stats` record of THIS plan (bench.py --launch-plan pinned)
        why = stale_reason(stats, digest, "the --stats record of %s" % key)
        roofline["rocprof_stats"] = None if why else stats
        if why:
            roofline["rocprof_stats_dropped"] = why
    if streams:
        roofline["device_streaming"] = streams
    if fused and per_launch >= 2:
        roofline["steps_per_launch"] = per_launch
        roofline["one_step_per_launch_equivalent"] = {"bytes_per_point_step": 4 * real, "achieved": per_launch * achieved, "frac": per_launch * achieved / HBM_PEAK_GBS,
                                                      "note": "bytes a one-step-per-launch kernel would have to move for the same work, over this launch's time"}
    if fused and per_launch == 3:
        # The three-step launch moves 2/3 of the bytes per grid-point-step of the two-step one: its fraction of the HBM roof is LOWER at a
        # HIGHER rate.  For comparison with earlier rounds' figure (two steps per launch, 16 B per grid-point-step in fp64):
        roofline["two_steps_per_launch_equivalent"] = {"bytes_per_point_step": 2 * real, "achieved": 1.5 * achieved, "frac": 1.5 * achieved / HBM_PEAK_GBS,
                                                       "frac_wall": 1.5 * achieved_wall / HBM_PEAK_GBS,
                                                       "note": "bytes the two-steps-per-launch kernel (rounds 4-5) moves for the same work, over this launch's time"}
    if fused:
        eq = rate(REALS_PER_POINT_STEP * real * pts_launch * per_launch, kernel_ms)
        roofline["survey_8d_equivalent"] = {"bytes_per_point_step": REALS_PER_POINT_STEP * real, "achieved": eq, "frac": eq / HBM_PEAK_GBS,
                                            "note": "staged-scheme traffic the same work would need (32 reals per grid-point-step); > peak because the launch does not move those bytes"}
    out = {"metric": "grid-point-steps/sec, %s torus %dx%d RK4" % ("FHN" if args.model == "fhn" else "Goldbeter", n, n), "value": value, "unit": "grid-point-steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.precision, "data": "synthetic (reference initial-condition rule: stable state + perturbed rectangle)",
           "config": {"workload": "%s_torus_%dx%d_%s_rk4" % (args.model, n, n, args.precision), "dt": dt, "stepper": "fused" if fused else "staged",
                      "decomposition": "phi-slabs x%d" % world, "L": 80.0, "W": 20.0, "D": 0.12, "beta": beta, "t_boundary": args.t_boundary, "halo": comm, "launch_plan": plan,
                      "preheat": preheat},
           "roofline": roofline}
    if staged:
        s_el, s_ms, s_kms, s_kernel, s_pts = staged
        s_alg, sb, s_step = REALS_PER_POINT_STAGE23 * real * s_pts, REALS_PER_POINT_STEP * real * n * n, s_el * 1e3 / args.staged_steps
        out["staged"] = {"steps": args.staged_steps, "ms_per_step": s_step, "value": n * n * args.staged_steps / s_el, "kernel": s_kernel + " (stage 2)", "kernel_ms": s_kms,
                         "algorithmic_bytes_per_launch": s_alg, "achieved": rate(s_alg, s_kms), "frac": rate(s_alg, s_kms) / HBM_PEAK_GBS,
                         "step_bytes_per_point": REALS_PER_POINT_STEP * real, "step_achieved": rate(sb, s_step), "step_frac": rate(sb, s_step) / HBM_PEAK_GBS,
                         "note": "SURVEY 8(d) accounting on the stepper it describes: four stage kernels per step, timed in this same process before the main run"}
    if per_rank:
        out["per_rank"] = per_rank
    return out


def run_local(args, crd, ndev, emit):
    """LOCAL LEG: every slab of the N-GPU run in THIS process as a LOCAL group (crd_comm_attach_local), slab k on device k: halos are
    device-to-device peer copies, one issuing host thread per device (crd_group_step_rk4_timed: event pairs on every slab's stream)."""
    n, world = args.size, args.gpus
    devices = [int(v) for v in args.devices.split(",")] if args.devices else list(range(world))
    if len(devices) != world or (ndev and max(devices) >= ndev):
        raise SystemExit("--transport local --gpus %d: need %d device ordinals below %d (--devices)" % (world, world, ndev))
    beta, params, dt, cfg = problem(args, crd)
    grp = crd.LocalGroup(params, world, devices=devices)
    try:
        grp.set_stepper(args.stepper)
        if args.issuing_threads and hasattr(grp, "set_threads"):
            grp.set_threads(args.issuing_threads)
        grp.set_exchange_period(args.exchange_period or (10 if min(s.nyl for s in grp.slabs) >= 256 else 8))
        grp.upload(crd.initial_conditions(cfg))
        for s in grp.slabs:
            if args.launch_plan:
                s.set_launch_plan(*[int(v) for v in args.launch_plan.split(",")])
            s.plan_launches()
        if not args.exchange_period and all(s.launch_plan().get("steps_per_launch", 1) == 3 for s in grp.slabs):
            grp.set_exchange_period(9)  # (plans that step triples inside the cycles -- fp32 -- want a period they divide: see the ring's rehearsal)
        preheat = {"ms_requested": args.preheat_ms, "steps": 0}
        if args.preheat_ms > 0:
            t0 = time.perf_counter()
            grp.step_rk4(0.0, dt, 8)
            est_ms = max((time.perf_counter() - t0) * 1e3 / 8, 0.01)
            n_pre = int(min(5000, max(0, args.preheat_ms / est_ms)))
            grp.step_rk4(0.0, dt, n_pre)
            preheat.update({"steps": 8 + n_pre, "ms": (time.perf_counter() - t0) * 1e3})
        grp.step_rk4(0.0, dt, args.warmup)  # (returns when every slab's last step is done)
        t0 = time.perf_counter()
        timings = grp.step_rk4_timed(args.warmup * dt, dt, args.steps)  # one dict per slab (crd_step_timing)
        elapsed = time.perf_counter() - t0
        peak = max(s.max_abs() for s in grp.slabs)
        if not (np.isfinite(peak) and peak <= 1e3):
            raise SystemExit("solution blew up (max|u| = %r): dt too large?" % peak)
        lead, tm0 = grp.slabs[0], timings[0]
        fused = lead.dominant_kernel().startswith("crd_rk4_fused")
        plan = lead.launch_plan()
        if tm0["kernel_ms"] > 0:
            kernel_ms, per_launch, pts = tm0["kernel_ms"], tm0.get("timed_steps_per_launch") or 1, n * lead.dominant_kernel_rows()
            source = "HIP events around one full-height launch on slab 0's compute stream inside the timed region (crd_group_step_rk4_timed)"
        else:  # nothing timed (a run shorter than an exchange cycle): the wall time of a LAUNCH, exchanges and gaps included
            per_launch = int(plan.get("steps_per_launch", 1)) if fused else 1
            kernel_ms, pts = elapsed * 1e3 / args.steps * per_launch / (1 if fused else 4), n * lead.nyl
            source = "wall clock per launch of the whole group (no launch of this short run was event-timed; exchanges and launch gaps included)"
        comm = {"transport": "local", "rccl_comm_count": None, "control_plane": None, "devices": devices, "exchange_period": {"steps": lead.exchange_period()} if fused else None,
                "issuing_threads": args.issuing_threads or "one per device (crd_group_step_rk4_timed)"}
        per_rank = [{"rank": k, "rows": s.nyl, "device": devices[k], "kernel_ms": timings[k]["kernel_ms"], "device_ms_per_step": timings[k]["ms_total"] / args.steps,
                     "launch_plan": s.launch_plan()} for k, s in enumerate(grp.slabs)]
        geometry = lead.launch_geometry() if fused and hasattr(lead, "launch_geometry") else None
        emit(json.dumps(build_line(args, crd, world, dt, beta, elapsed, kernel=lead.dominant_kernel(), kernel_ms=kernel_ms, launches=1 if fused else 2, pts_launch=pts,
                                   device_ms_per_step=max(t["ms_total"] for t in timings) / args.steps, plan=plan, comm=comm, preheat=preheat, staged=None, per_rank=per_rank,
                                   kernel_ms_source=source, per_launch=per_launch, geometry=geometry)))
    finally:
        grp.close()


def run(args, crd, world, rank, local_rank, device_sync, emit, t_start=None):
    """RANK: the benchmark proper on top of the `crd` API (tests/test_distributed_cpu.py drives this very function with 2, 3 and 8
    processes over gloo against a stand-in whose halos travel by the library's own ring plan).  device_sync(): wait for the device."""
    t_start = time.monotonic() if t_start is None else t_start
    ctl = ControlPlane(world, rank)
    n = args.size
    beta, params, dt, cfg = problem(args, crd)
    slab = None

    def bail(msg):
        """Leave with a non-zero status from a clean state: every rank takes the same exit, context and process group torn down first."""
        if slab is not None:
            slab.close()
        ctl.close()
        raise SystemExit(msg)

    def fence():
        ctl.barrier()
        device_sync()

    # ROLL CALL: every rank reports whether its own set-up worked BEFORE anybody enters a collective of the ring -- a rank that failed
    # alone (no device memory, RCCL library missing) would otherwise leave the others waiting inside ncclCommInitRank.
    trouble = ""
    try:
        slab = crd.Slab(params, rank, world, local_rank)
        ident = crd.rccl_unique_id() if (rank == 0 and (world > 1 or args.force_rccl)) else b""
    except Exception as e:  # noqa: BLE001 -- reported through the control plane, then every rank leaves
        trouble, ident = "rank %d: %s" % (rank, e), b""
    failed = [p for p in ctl.gather(trouble) if p]
    if failed:
        bail("set-up failed: " + "; ".join(failed))
    # RING BRING-UP in a helper thread, before anything is timed: a ring that hangs at first contact cannot be interrupted from Python,
    # but the main thread can stop waiting for it.  Then the AGREED OUTCOME over the control plane: all take the same way on.
    ring_wanted = world > 1 or args.force_rccl
    ident = ctl.broadcast_bytes(ident, 128) if world > 1 else ident
    up = {"status": "ok", "bad": 0, "info": None}

    def bring_ring_up():
        try:
            if ring_wanted:
                slab.init_rccl(ident)
            up["info"] = slab.comm_info()
            if up["info"][0] == "rccl":
                up["depth"] = min(32, slab.grid.ny // world)  # the same on every rank; 32 = what the deep-halo cycle exchanges
                up["bad"] = halo_selfcheck(crd, slab, rank, world, up["depth"])
        except Exception as e:  # noqa: BLE001 -- reported through the control plane
            up["status"] = "rank %d: %s" % (rank, e)

    stuck = False
    if ring_wanted:
        import threading

        th = threading.Thread(target=bring_ring_up, daemon=True)
        th.start()
        th.join(args.ring_timeout_s if args.ring_timeout_s > 0 else None)
        stuck = th.is_alive()
        if stuck:
            up["status"] = "rank %d: the RCCL ring did not come up within %.0f s (communicator set-up / first exchange)" % (rank, args.ring_timeout_s)
            sys.stderr.write("bench.py: %s\n" % up["status"])
            sys.stderr.flush()
    else:
        bring_ring_up()
    failures = [st for st in ctl.gather(up["status"]) if st != "ok"]
    if failures:
        ring_failed(args, ctl, slab, rank, world, failures, stuck, emit, t_start)  # does not return
    slab.set_stepper(args.stepper)
    transport, comm_ranks, comm_rank = up["info"]
    comm = {"transport": transport, "rccl_comm_count": comm_ranks if transport == "rccl" else None, "control_plane": "gloo" if world > 1 else None}
    if transport == "rccl" and os.environ.get("CRD_RCCL_LIBRARY"):
        comm["rccl_library_override"] = os.environ["CRD_RCCL_LIBRARY"]  # not librccl (the tests' shared-memory stand-in): a rehearsal of the control flow, never a figure
    if transport == "rccl":
        bad_total, wrong_count, wrong_rank = ctl.sum_ints([up["bad"], int(comm_ranks != world), int(comm_rank != rank)])
        comm["halo_selfcheck"] = {"depth": up["depth"], "fields": 2, "ghost_rows_checked_per_rank": 4 * up["depth"], "mismatching_values": bad_total,
                                  "ok": bad_total == 0 and wrong_count == 0 and wrong_rank == 0}
        if not comm["halo_selfcheck"]["ok"]:
            if args.preflight and rank == 0:
                emit(json.dumps({"preflight": True, "ok": False, "n_gpus": world, "halo": comm, "elapsed_s": round(time.monotonic() - t_start, 2),
                                 "reasons": ["halo self-check: %d ghost values differ; %d ranks see a communicator of the wrong size, %d the wrong rank" % (bad_total, wrong_count, wrong_rank)]}))
            bail("halo self-check failed: %d ghost values differ from the neighbours' rows; %d ranks see a communicator of the wrong size, %d the wrong rank"
                 % (bad_total, wrong_count, wrong_rank))
    if args.preflight:
        # PREFLIGHT ends here: every rank answered the roll call, the ring came up, RCCL counts `world` ranks, 32 ghost rows of both fields
        # arrived from the right neighbours.  Nothing was planned, stepped or timed.
        if rank == 0:
            emit(json.dumps({"preflight": True, "ok": True, "n_gpus": world, "halo": comm, "elapsed_s": round(time.monotonic() - t_start, 2),
                             "kernel_table_digest": crd.kernel_table_digest() if hasattr(crd, "kernel_table_digest") else None}))
        slab.close()
        ctl.close()
        return
    y_init = crd.initial_conditions(cfg, slab.js, slab.je)
    slab.upload(y_init)

    # Beside a fused run (N = 1): the staged stepper -- the scheme SURVEY 8(d)'s 256 B per grid-point-step describes -- in the same process.
    staged = None
    will_fuse = args.stepper != "staged" and slab.dominant_kernel().startswith("crd_rk4_fused")
    if world == 1 and will_fuse and args.staged_steps > 0:
        slab.set_stepper("staged")
        slab.step_rk4(0.0, dt, 4, sync=True)
        fence()
        t0 = time.perf_counter()
        s_ms, s_kms, _ = slab.step_rk4_timed(4 * dt, dt, args.staged_steps)
        fence()
        staged = (time.perf_counter() - t0, s_ms, s_kms, slab.dominant_kernel(), n * slab.dominant_kernel_rows())
        slab.set_stepper(args.stepper)
        slab.upload(y_init)
    if args.launch_plan:
        slab.set_launch_plan(*[int(v) for v in args.launch_plan.split(",")])
    t_plan = time.perf_counter()
    slab.plan_launches()  # the launch plan is measured here, outside every timed region (also with --warmup 0)
    t_plan = time.perf_counter() - t_plan

    # REHEARSALS (ring runs with the one-launch stepper).  Slack: three exchange cycles with per-exchange event pairs; if on ANY rank the
    # compute stream stood at its wait for a halo for longer than a queue latency, every rank gives the exchange a third sweep of cover
    # (crd_set_halo_slack; same bits).  Period: 8, 10 and 16 steps per exchange, 80 steps each (whole cycles of all three), twice; the fastest
    # (MAX over ranks of each rank's best) is kept, a longer period than 8 only if it wins by > 1 %.  Both go to the top level of config.halo.
    if transport == "rccl" and will_fuse:
        def rehearse():
            slab.set_diagnostics(True)
            slab.step_rk4_timed(0.0, dt, 24)
            tm = slab.step_timing()
            slab.set_diagnostics(False)
            return ctl.max_float(tm["exposed_halo_ms"] / max(1, tm["halo_waits"]))

        exposed, sweeps = [rehearse()], 1
        if exposed[0] > args.halo_slack_threshold_ms:
            sweeps = 2
            slab.set_halo_slack(2)
            exposed.append(rehearse())
        comm["slack"] = {"sweeps": sweeps, "threshold_ms": args.halo_slack_threshold_ms, "rehearsal_exposed_halo_ms_max_over_ranks": exposed}
        period = {"steps": slab.exchange_period(), "chosen_by": "default"}
        if args.exchange_period:
            slab.set_exchange_period(args.exchange_period)
            period = {"steps": args.exchange_period, "chosen_by": "--exchange-period"}
        elif min(crd.slab_extents(slab.grid.ny, k, world)[1] - crd.slab_extents(slab.grid.ny, k, world)[0] + 1 for k in range(world)) >= 256:
            trial = {}
            # (a plan that steps triples inside the cycles -- fp32 -- wants a period it divides: 8 = 3 + 3 + 2 and 10 = 3 + 3 + 3 + 1 end on
            # another kernel; measured on a rank's 16384 x 2048 share: 9 and 12 steps 58.9 / 59.0 us, 8 / 10 / 16 64.0 / 63.4 / 61.3)
            triples = ctl.sum_ints([int(slab.launch_plan().get("steps_per_launch", 1) == 3)])[0] == world
            periods, span = ((8, 9, 12, 16), 144) if triples else ((8, 10, 16), 80)  # (span: whole cycles of each)
            for e in periods:
                slab.set_exchange_period(e)
                slab.step_rk4_timed(0.0, dt, 2 * e)  # (settle into the cycle)
                trial[e] = ctl.max_float(min(slab.step_rk4_timed(0.0, dt, span)[0] / span for _ in range(2)))
            keep = min(trial, key=lambda e: trial[e] * (1.0 if e == 8 else 1.01))  # (a longer period has to win by 1 %)
            slab.set_exchange_period(keep)
            period = {"steps": keep, "chosen_by": "rehearsal", "rehearsal_device_ms_per_step_max_over_ranks": {str(k): v for k, v in trial.items()}}
        comm["exchange_period"] = period
        slab.upload(y_init)
    del y_init

    # Pre-heat (the same 20 steps measured 5 % slower behind 5 warm-up steps than behind 800: profiles/r03/warm_clocks.txt), warm-up, then
    # the timed region: barrier + device synchronisation, K steps, device synchronisation, time stamp, barrier; MAX over the ranks.
    preheat = {"ms_requested": args.preheat_ms, "steps": 0}
    if args.preheat_ms > 0:
        t0 = time.perf_counter()
        slab.step_rk4(0.0, dt, 8, sync=True)
        est_ms = max((time.perf_counter() - t0) * 1e3 / 8, 0.01)
        n_pre = int(min(5000, max(0, ctl.max_float(args.preheat_ms / est_ms))))
        slab.step_rk4(0.0, dt, n_pre, sync=True)
        preheat.update({"steps": 8 + n_pre, "ms": (time.perf_counter() - t0) * 1e3})
    slab.step_rk4(0.0, dt, args.warmup, sync=True)
    fence()
    t0 = time.perf_counter()
    ms_dev, kernel_ms, launches = slab.step_rk4_timed(args.warmup * dt, dt, args.steps)
    device_sync()
    elapsed_rank = time.perf_counter() - t0
    ctl.barrier()
    elapsed = ctl.max_float(elapsed_rank)
    timed = slab.step_timing()
    spread = []  # the same K steps again, a few times, same fences: the spread of the figure on this box (never part of `value`)
    for r in range(max(0, args.repeats)):
        fence()
        t0 = time.perf_counter()
        slab.step_rk4((args.warmup + (r + 1) * args.steps) * dt, dt, args.steps, sync=True)
        device_sync()
        spread.append(ctl.max_float(time.perf_counter() - t0) * 1e3 / args.steps)
    plan_used = dict(slab.launch_plan(), measured_in_s=round(t_plan, 3))
    geometry = slab.launch_geometry() if will_fuse and hasattr(slab, "launch_geometry") else None
    # Beside a two-steps-per-launch run (N = 1): the one-step-per-launch kernel, which crosses memory once per step, in this same process.
    one_step = None
    if world == 1 and will_fuse and plan_used.get("steps_per_launch", 1) >= 2 and not args.launch_plan and args.one_step_steps > 0:
        slab.set_launch_plan(0, plan_used["xcd_mapping"], plan_used["columns_per_lane"], 1, 1)
        slab.step_rk4(0.0, dt, 8, sync=True)
        o_ms, o_kms, _ = slab.step_rk4_timed(0.0, dt, args.one_step_steps)
        one_step = (o_ms / args.one_step_steps, o_kms, slab.launch_plan())
    peak = slab.max_abs()
    blown = ctl.sum_ints([0 if (np.isfinite(peak) and peak <= 1e3) else 1])[0]
    if blown:
        bail("solution blew up on %d rank(s) (this rank: max|u| = %r): dt too large?" % (blown, peak))

    # Per-rank figures (ring runs): the timed region's own, then a short pass with per-exchange event pairs -- three cycles, outside the rate.
    per_rank = None
    if transport == "rccl":
        rec = {"rank": rank, "rows": slab.nyl, "kernel_ms": kernel_ms, "kernel_rows": slab.dominant_kernel_rows(), "ms_per_step": elapsed_rank * 1e3 / args.steps,
               "device_ms_per_step": ms_dev / args.steps, "launch_plan": slab.launch_plan()}
        diag_steps = 3 * slab.exchange_period()
        slab.set_diagnostics(True)
        d_ms, _, _ = slab.step_rk4_timed((args.warmup + args.steps) * dt, dt, diag_steps)
        tm = slab.step_timing()
        slab.set_diagnostics(False)
        rec.update({"diag_steps": diag_steps, "diag_ms_per_step": d_ms / diag_steps, "halo_waits": tm["halo_waits"], "exchanges": tm["exchanges"],
                    "exposed_halo_ms": tm["exposed_halo_ms"] / max(1, tm["halo_waits"]), "exchange_ms": tm["exchange_ms"] / max(1, tm["exchanges"]),
                    "halo_slack": tm["halo_slack"], "agreement_restarts": tm["agreement_restarts"]})
        per_rank = ctl.gather(rec)
        # did RCCL see N ranks, and did the exchange hide?  At a glance, at the top level of config.halo:
        comm["exposed_halo_ms_per_rank"] = [q["exposed_halo_ms"] for q in per_rank]
        comm["exchange_ms_per_rank"] = [q["exchange_ms"] for q in per_rank]

    if rank == 0:
        out = build_line(args, crd, world, dt, beta, elapsed, kernel=slab.dominant_kernel(), kernel_ms=kernel_ms, launches=launches,
                         pts_launch=n * (per_rank[0]["kernel_rows"] if per_rank else slab.dominant_kernel_rows()),  # (the rows of the launch the timed region timed)
                         device_ms_per_step=ms_dev / args.steps, plan=plan_used, comm=comm, preheat=preheat, staged=staged, per_rank=per_rank,
                         kernel_ms_source="HIP events around sampled launches on the library's compute stream inside the timed region",
                         per_launch=timed.get("timed_steps_per_launch") or None, geometry=geometry)
        if per_rank:
            out["per_rank_note"] = ("kernel_ms / ms_per_step / device_ms_per_step: each rank's own figures from the timed region; exposed_halo_ms: average time its compute "
                                    "stream stood at the wait for a halo, per exchange, and exchange_ms: average duration of an exchange on the second stream -- both from a "
                                    "%d-step diagnostic pass after the timed region" % per_rank[0].get("diag_steps", 0))
        if one_step:
            o_rate = 4 * (8 if args.precision == "f64" else 4) * n * n / (one_step[1] * 1e-3) / 1e9
            out["roofline"]["one_step_per_launch"] = {"kernel_ms": one_step[1], "device_ms_per_step": one_step[0], "steps": args.one_step_steps, "achieved": o_rate,
                                                      "frac": o_rate / HBM_PEAK_GBS, "plan_key": crd.plan_key(args.model, args.precision, one_step[2]),
                                                      "note": "the same grid stepped with ONE step per launch, timed after the main run in this process: read + write of both fields once per step"}
        if spread:
            out["timing_spread"] = {"repeats_ms_per_step": spread, "min": min(spread), "max": max(spread),
                                    "note": "the K timed steps repeated %d times after the timed region, same fences; `value` is the first, timed pass only" % len(spread)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, beta, dt)
        emit(json.dumps(out))
    slab.close()
    ctl.close()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:  # noqa: BLE001 -- say which rank it was; the launcher ends the other ranks when this one exits non-zero
        sys.stderr.write("bench.py: rank %s failed\n%s" % (os.environ.get("RANK", "0"), traceback.format_exc()))
        sys.stderr.flush()
        sys.exit(1)
