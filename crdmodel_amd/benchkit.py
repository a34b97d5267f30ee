"""benchkit.py -- the parts of bench.py that are not the benchmark: what the ranks of a run tell each other (a gloo group on CPU tensors),
the halo self-check (ghost rows == the rows the ring neighbours own, through the library's own transport), and the committed
profiler records a bench line quotes.  No GPU work and no timing in here; bench.py re-exports these names."""
import datetime
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (16 on a 1-GPU box)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def committed_json(name):
    """profiles/<name> as a dict ({} when absent): measurements this run does not repeat but quotes, with their provenance."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}


def stale_reason(rec, kernel_digest, what):
    """Why a committed profile record must not be quoted for the kernel this run launched ("" when it may): records carry the digest of
    the kernel they were measured on (crdmodel_amd.kernel_digest: registers and loop instruction mix as the assembler printed them)."""
    if kernel_digest is None or not rec:
        return ""  # (callers without a kernel to compare with: the tests' stand-in)
    have = rec.get("kernel_digest", "")
    if have == kernel_digest:
        return ""
    return "%s was measured on another build of this kernel (record %s, loaded library %s): not quoted" % (what, have or "unstamped", kernel_digest or "no kernel table")


def measured_traffic(kernel_key, points, kernel_digest=None):
    """(HBM bytes per launch, provenance): the committed result of separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
    with this plan pinned (profiles/pmc_traffic.json: bytes per grid point), scaled by the points one launch of this run covers --
    counters need rocprofv3 around the process, the benchmark cannot collect them itself.  (None, reason) when there is no record, or
    when the record was measured on another build of the kernel (kernel_digest: see stale_reason)."""
    rec = committed_json("pmc_traffic.json").get(kernel_key)
    if not rec:
        return None, "no rocprofv3 --pmc passes recorded for %s" % kernel_key
    why = stale_reason(rec, kernel_digest, "the --pmc record of %s (%s)" % (kernel_key, rec.get("source", "profiles/pmc_traffic.json")))
    if why:
        return None, why
    return rec["bytes_per_point"] * points, "%s: %.2f B/point on %s, rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE in separate passes; not re-measured by this run" % (
        rec.get("source", "profiles/pmc_traffic.json"), rec["bytes_per_point"], rec.get("grid", "?"))


def selfcheck_pattern(rows, nx, var, dtype):
    """The self-check field on global rows `rows`: f(global row, column, variable), exactly representable in fp32 too."""
    rows = np.asarray(rows, dtype=np.int64)
    return (((rows[:, None] * 7919 + np.arange(nx, dtype=np.int64)[None, :] * 31 + var * 5) % 16777213).astype(np.float64)).astype(dtype)


def selfcheck_ghost_rows(js, je, ny, depth):
    """Global rows a slab [js, je] of a periodic ny-row grid must find in its low / high ghost rows after an exchange of `depth`."""
    return np.arange(js - depth, js) % ny, np.arange(je + 1, je + 1 + depth) % ny


def halo_selfcheck(crd, slab, rank, world, depth=32):
    """Did the transport move the right rows?  Every rank uploads the self-check field, runs ONE exchange of `depth` ghost rows through
    the library's own transport and compares its ghost rows with the rows its ring neighbours own.  Returns the mismatches (0 = ok)."""
    nx, nyl, ny = slab.nx, slab.nyl, slab.grid.ny
    own = np.arange(slab.js, slab.je + 1)
    y = np.empty((nyl, nx, 2), dtype=np.float64)
    y[..., 0], y[..., 1] = selfcheck_pattern(own, nx, 0, slab.dtype), selfcheck_pattern(own, nx, 1, slab.dtype)
    slab.upload(y)
    slab.halo_exchange(depth)
    lo_rows, hi_rows = selfcheck_ghost_rows(slab.js, slab.je, ny, depth)
    bad = 0
    for var in (0, 1):
        bad += int(np.count_nonzero(slab.download_rows(var, -depth, depth) != selfcheck_pattern(lo_rows, nx, var, slab.dtype)))
        bad += int(np.count_nonzero(slab.download_rows(var, nyl, depth) != selfcheck_pattern(hi_rows, nx, var, slab.dtype)))
    return bad


class ControlPlane:
    """Everything the ranks tell each other outside the halo exchange, over a gloo group on CPU tensors (a second, torch-owned RCCL
    communicator would be one more thing to go wrong on first contact with an 8-GPU node).  world == 1: no group at all."""

    def __init__(self, world, rank, timeout_s=300):
        self.world, self.rank, self.dist = world, rank, None
        if world > 1:
            import torch.distributed as dist

            if int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) == world:  # one node: loopback is always there, the hostname may not resolve
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def _reduce(self, values, dtype_name, op):
        if not self.dist:  # one process: nothing to reduce, and no torch needed (the stand-in path of the tests)
            return list(values)
        import torch

        t = torch.tensor(list(values), dtype=getattr(torch, dtype_name))
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.tolist()

    def sum_ints(self, values):
        return [int(v) for v in self._reduce(values, "int64", "SUM")]

    def max_float(self, value):
        return float(self._reduce([value], "float64", "MAX")[0])

    def broadcast_bytes(self, payload, nbytes, src=0):
        if not self.dist:
            return payload
        import torch

        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.rank == src:
            t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
        self.dist.broadcast(t, src=src)
        return bytes(t.numpy().tobytes())

    def gather(self, obj):
        """[rank 0's obj, rank 1's obj, ...] on every rank."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.destroy_process_group()
        self.dist = None
