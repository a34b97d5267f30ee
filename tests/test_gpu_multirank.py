"""The ring path with MORE THAN ONE RANK on hardware: WORLD processes, each with one slab of the grid on the one GPU of the box,
stepping through the product's RCCL code path -- deep-halo exchange cycles, the ring-wide agreement on the cycle position,
two-steps-per-launch plans that differ from rank to rank, an upload on one rank between calls, the error-controlled integrators'
norm all-reduce and resume vote.  RCCL itself refuses two ranks on one device, so the bytes travel through the stand-in of
tests/native/ring_standin_rccl.cpp (bound through CRD_RCCL_LIBRARY; test infrastructure, see its header); everything above the
eleven nccl* entry points is the shipped library.  The states are compared with a single periodic slab running the same
programme: bit for bit after fixed steps, to round-off after an error-controlled integration (the norm is summed in another
order) that took the same steps."""
import ctypes
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import crdmodel_amd as crd
import ring_rank_worker as worker

pytestmark = pytest.mark.gpu

NATIVE = os.path.join(ROOT, "tests", "native")
STANDIN = os.path.join(NATIVE, "_build", "libring_standin_rccl.so")


def build_standin():
    src = os.path.join(NATIVE, "ring_standin_rccl.cpp")
    if not os.path.exists(STANDIN) or os.path.getmtime(STANDIN) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        os.makedirs(os.path.dirname(STANDIN), exist_ok=True)
        subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", STANDIN, "-lpthread", "-lrt"], check=True, capture_output=True)
    return STANDIN


@pytest.fixture(scope="module")
def standin():
    return build_standin()


def unique_id(standin):
    lib = ctypes.CDLL(standin)
    buf = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(buf) == 0
    return buf.raw


def run_ring(standin, spec, world, tmp_path, shuffle_seed=None):
    spec_path = tmp_path / "programme.json"
    spec_path.write_text(json.dumps(spec))
    raw_id = unique_id(standin)
    ident = raw_id.hex()
    env = dict(os.environ, CRD_RCCL_LIBRARY=standin, CRD_STANDIN_TIMEOUT_S="40")
    if shuffle_seed is not None:  # the operations of every ncclGroup complete in a random order, each behind a random latency of up to 300 us
        env.update(CRD_STANDIN_SHUFFLE=str(shuffle_seed), CRD_STANDIN_LATENCY_US="300")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ring_rank_worker.py"), str(r), str(world), ident, str(spec_path), str(tmp_path / ("rank%d.npz" % r))],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            try:
                outs.append(pr.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
    finally:
        # a rank that died before ncclCommDestroy leaves the ring's shared-memory segment behind (it lives in memory): remove it
        segment = os.path.join("/dev/shm", raw_id.split(b"\0", 1)[0].decode().lstrip("/"))
        if os.path.exists(segment):
            os.unlink(segment)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(o[-3000:] for o in outs)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    n_shots = len([k for k in parts[0].files if k.startswith("shot")])
    shots = [np.concatenate([part["shot%d" % k] for part in parts]) for k in range(n_shots)]
    return shots, [part["stats"] for part in parts]


def run_single(spec, world):
    p = worker.problem(crd, spec)
    with crd.Slab(p) as one:
        one.upload(crd.initial_conditions(crd.run_config(p, wave_length=0.1, wave_width=0.5)))
        return worker.run_programme(crd, one, spec, None, world)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def plans(world, *per_rank):
    """world + 1 plans: ranks 0 .. world-1, then the single slab's."""
    return [list(per_rank[k % len(per_rank)]) for k in range(world + 1)]


def test_triples_inside_the_exchange_cycles_of_an_fp32_ring(gpu_device, standin, tmp_path):
    """Round 6: fp32 slabs step THREE steps per launch inside their exchange cycles (a strip per wavefront, two columns per lane), each
    rank by its own plan -- triples beside pairs beside single steps.  Two cases the soaks found when the ring first met triples:
    (1) a call that starts E - 3 steps into a cycle begins with a triple that is the cycle's LAST launch, i.e. with an exchange --
    it must not be issued ahead of the ring's agreement on the cycle position (here: an upload on one rank just before, so the ring
    does disagree and starts afresh; the send / receive sequences stopped pairing: tools/soak_ring_processes.py, seed 70);
    (2) at period 3 a triple from the cycle's start would be its first launch (which waits for the halo) and its last (which sends
    the next) at once: it is stepped as a pair and a single step (tests/long_oracle_sweep.py, seed 71).  Bit-equal to the single slab."""
    world = 3
    ny = world * 131 + 2
    spec = dict(model="fhn", surface="torus", nx=208, ny=ny, precision="f32", t_boundary=0.0, dt_factor=0.7, vary_beta=1)
    dt = spec["dt_factor"] * crd.stable_dt(worker.problem(crd, spec))
    spec["t_boundary"] = 30.4 * dt
    three, two, one = (0, 0, 2, 1, 3), (0, 1, 2, 1, 2), (2, 1, 2, 0, 1)
    spec["programme"] = [
        ["plan", plans(world, three, three, one)], ["step", 5], ["scale_rows_of", 1, 1.0009765625], ["step", 9], ["snapshot"],   # period 8: q0 = 5 = E - 3
        ["step", 13], ["scale_rows_of", 0, 0.9990234375], ["step", 4], ["snapshot"],
        ["period", 3], ["step", 10], ["snapshot"], ["period", 4], ["step", 7], ["scale_rows_of", 2, 1.0009765625], ["step", 9], ["snapshot"],
        ["period", 9], ["plan", plans(world, three, two, three)], ["slack", 2], ["step", 6], ["scale_rows_of", 1, 0.9990234375], ["step", 21], ["timed", 12],
    ]
    ring, _ = run_ring(standin, spec, world, tmp_path, shuffle_seed=3)
    single, _ = run_single(spec, world)
    assert len(ring) == len(single) == 5
    for k, (a, b) in enumerate(zip(ring, single)):
        assert np.array_equal(a, b), "snapshot %d: %.3e" % (k, rel(a, b))
    assert np.abs(single[-1] - single[0]).max() > 1e-3


@pytest.mark.parametrize("world,shuffle_seed", [(2, None), (3, None), (4, None), (2, 6), (5, 11)], ids=["2", "3", "4", "2-shuffled", "5-shuffled"])
def test_fixed_steps_on_a_ring_of_processes(gpu_device, standin, tmp_path, world, shuffle_seed):
    """Exchange cycles carried across calls, periods 5 / 8 / 16, halo slack 2, launch plans that pair steps on some ranks only,
    absorbing rows on rank 0 / the last rank (tBoundary inside the run), a one-rank upload in mid-cycle (the ring then agrees to
    start a fresh cycle), staged <-> one-launch stepper switches: the ring's rows == the single slab's, bit for bit.
    "shuffled" (round 6): the transport completes the operations of every ncclGroup in a random order behind random latencies
    (tests/native/ring_standin_rccl.cpp: run_group_shuffled) -- at two ranks both neighbours are ONE peer and the two rows a rank sends it
    are told apart by nothing but their order of issue (crd_halo_plan), the case the reference's Exchange() leaves to MPI's message
    ordering (src/FHNmodel_torus.cpp:805,811); five ranks are the most this box's six-process limit on the GPU allows."""
    ny = world * 131 + (world - 1)  # slabs of unequal height, every one above the 64 ghost rows of period 16 and above 4 bands of 32
    spec = dict(model="fhn", surface="torus", nx=200, ny=ny, precision="f64", t_boundary=0.0, dt_factor=0.7, vary_beta=1)
    dt = spec["dt_factor"] * crd.stable_dt(worker.problem(crd, spec))
    spec["t_boundary"] = 21.3 * dt
    one_step, two_steps = (0, 0, 1, 1, 1), (0, 1, 1, 1, 2)
    spec["programme"] = [
        ["step", 5], ["snapshot"], ["step", 13], ["timed", 9], ["snapshot"],
        ["period", 5], ["step", 11], ["snapshot"],
        ["plan", plans(world, two_steps, one_step)], ["step", 17], ["snapshot"],
        ["slack", 2], ["step", 23], ["scale_rows_of", world - 1, 1.0009765625], ["step", 7], ["snapshot"],
        ["stepper", "staged"], ["step", 3], ["stepper", "fused"], ["step", 6], ["snapshot"],
        ["period", 16], ["plan", plans(world, one_step, two_steps)], ["step", 35], ["scale_rows_of", 0, 0.9990234375], ["step", 2],
    ]
    ring, _ = run_ring(standin, spec, world, tmp_path, shuffle_seed)
    single, _ = run_single(spec, world)
    assert len(ring) == len(single) == 7
    for k, (a, b) in enumerate(zip(ring, single)):
        assert np.array_equal(a, b), "snapshot %d: %.3e" % (k, rel(a, b))
    assert np.abs(single[-1] - single[0]).max() > 1e-3


@pytest.mark.parametrize("world,model,precision", [(2, "fhn", "f64"), (3, "fhn", "f64"), (2, "goldbeter", "f64")])
def test_error_controlled_integration_on_a_ring_of_processes(gpu_device, standin, tmp_path, world, model, precision):
    """Both integrators across processes: the attempts' error norms go through the all-reduce, every rank takes the same decisions
    (same accepted / rejected counts and last step on every rank; the single slab's too, as long as no last-bit difference of a
    norm -- summed in another order there -- tips a decision), a call resumes the previous one's
    internal state only if EVERY rank can (the upload on one rank in between makes all of them start afresh), fixed steps and
    error-controlled calls alternate."""
    ny = world * 97 + 1
    spec = dict(model=model, surface="torus", nx=152, ny=ny, precision=precision, t_boundary=0.0, dt_factor=0.6)
    spec["programme"] = [
        ["adaptive", 1, 22.0, 1], ["snapshot"], ["adaptive", 1, 9.5, 1], ["snapshot"],
        ["step", 4], ["adaptive", 1, 7.0, 1], ["scale_rows_of", 0, 1.0009765625], ["adaptive", 1, 12.0, 1], ["snapshot"],
        ["adaptive", 0, 9.0, 0], ["adaptive", 0, 6.0, 1], ["adaptive", 0, 5.0, 1], ["step", 3],
    ]
    ring, ring_stats = run_ring(standin, spec, world, tmp_path)
    single, single_stats = run_single(spec, world)
    for st in ring_stats[1:]:
        assert np.array_equal(st, ring_stats[0]), (st, ring_stats[0])  # one norm for all: every rank takes the very same decisions
    assert single_stats[:, 0].min() >= 2
    # Against the single slab the norm is summed in another order: where the controller sits at a limit, a last-bit difference can
    # change a later step size, and the step sequences part (then the states agree to the integrator's tolerance, not to round-off).
    # (How far a last-bit difference goes: measured on the 3-rank case, round 5 -- one ulp in an attempt's step size moves that attempt's
    # error sum by 4e-10 relative, the estimate being a difference of O(h f) terms that cancel to O(h^5), and the step sizes of the
    # following attempts by 1e-11 ... 1e-8.  Same decisions and step sizes to 1e-6 is what "the same sequence" means here; the STATES
    # below are held to round-off.)
    same = np.array_equal(ring_stats[0][:, :2], single_stats[:, :2]) and np.allclose(ring_stats[0][:, 2], single_stats[:, 2], rtol=1e-6, atol=0.0)
    first_calls = ring_stats[0][:2], single_stats[:2]
    assert np.array_equal(first_calls[0][:, :2], first_calls[1][:, :2]) and np.allclose(first_calls[0][:, 2], first_calls[1][:, 2], rtol=1e-6, atol=0.0), first_calls
    for k, (a, b) in enumerate(zip(ring, single)):
        assert rel(a, b) <= (1e-9 if same or k < 2 else 1e-5), "snapshot %d: %.3e" % (k, rel(a, b))


def test_fp32_flat_ring_of_processes(gpu_device, standin, tmp_path):
    """fp32, flat surface, Goldbeter kinetics, two columns per lane on one rank and one on the other: bit-equal to the single slab."""
    world = 2
    spec = dict(model="goldbeter", surface="flat", nx=256, ny=2 * 140, precision="f32", t_boundary=0.0, dt_factor=0.5)
    spec["programme"] = [["step", 9], ["plan", plans(world, (0, 0, 2, 1, 2), (1, 1, 1, 0, 1))], ["step", 21], ["snapshot"], ["period", 3], ["step", 8]]
    ring, _ = run_ring(standin, spec, world, tmp_path)
    single, _ = run_single(spec, world)
    for a, b in zip(ring, single):
        assert np.array_equal(a, b)


def test_a_rank_that_stays_away_is_an_error_not_a_hang(gpu_device, standin, tmp_path):
    """The stand-in's own promise: a ring one of whose ranks never calls in ends with an error from the library on the others."""
    spec = dict(model="fhn", surface="torus", nx=64, ny=2 * 80, precision="f64", t_boundary=0.0, dt_factor=0.5, programme=[["step", 3]])
    spec_path = tmp_path / "programme.json"
    spec_path.write_text(json.dumps(spec))
    env = dict(os.environ, CRD_RCCL_LIBRARY=standin, CRD_STANDIN_TIMEOUT_S="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ring_rank_worker.py"), "0", "2", unique_id(standin).hex(), str(spec_path), str(tmp_path / "r0.npz")],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "not every rank called ncclCommInitRank" in (r.stdout + r.stderr)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_with_rank_processes_on_one_gpu(gpu_device, standin, world):
    """`python bench.py --gpus N` end to end with N REAL rank processes of the shipped library (the self-launcher, the gloo control
    plane, communicator set-up from a broadcast id, the halo self-check across ranks, the exchange-period and slack rehearsals, the
    per-rank gather, the JSON line) -- the ranks share the box's one GPU and the stand-in transport moves the halos, which the
    line says (`config.halo.rccl_library_override`): a rehearsal of the control flow, not a figure."""
    env = dict(os.environ, CRD_RCCL_LIBRARY=standin, CRD_STANDIN_TIMEOUT_S="60")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--transport", "rccl", "--size", "2048", "--steps", "24", "--warmup", "6",
                        "--repeats", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    halo = d["config"]["halo"]
    assert d["n_gpus"] == world and d["steps"] == 24 and d["value"] > 0 and d["scaling"] == "strong"
    assert halo["transport"] == "rccl" and halo["rccl_comm_count"] == world and halo["rccl_library_override"] == standin
    assert halo["halo_selfcheck"]["ok"] and halo["halo_selfcheck"]["mismatching_values"] == 0
    assert d["config"]["decomposition"] == "phi-slabs x%d" % world and len(d["per_rank"]) == world
    assert d["config"]["launcher"]["transports_tried"] == ["rccl"]


def test_ranks_under_an_external_launcher_fall_back_to_the_local_leg(gpu_device, standin):
    """Two rank processes of bench.py started the way torch.distributed.run starts them (RANK / WORLD_SIZE / MASTER_* set by the caller),
    with a ring that does not come up (the stand-in's ncclCommInitRank fails on request): the ranks settle it over the control plane,
    rank 1 leaves with status 0, rank 0 runs the LOCAL leg (both slabs on the box's one GPU: --devices 0,0) in a child process and
    passes its line on with the reason -- the shipped library and kernels throughout."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, CRD_RCCL_LIBRARY=standin, CRD_STANDIN_FAIL_INIT="1", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.pop("CRD_BENCH_SELF_LAUNCHED", None)
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--size", "2048", "--steps", "24", "--warmup", "6", "--repeats", "1",
            "--no-cpu-baseline"]
    procs = [subprocess.Popen(argv, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    res = []
    for pr in procs:
        try:
            o, e = pr.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        res.append((pr.returncode, o, e))
    assert [r[0] for r in res] == [0, 0], [(r[0], r[2][-2000:]) for r in res]
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not res[1][1].strip()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["halo"]["transport"] == "local" and d["config"]["halo"]["devices"] == [0, 0]
    la = d["config"]["launcher"]
    assert la["transports_tried"] == ["rccl", "local"] and "external launcher" in la["mode"] and "CRD_STANDIN_FAIL_INIT" in la["fallback_reasons"][0]
