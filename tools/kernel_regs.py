#!/usr/bin/env python3
"""Registers, occupancy and the instruction mix of the steady-state loop of every kernel in a .hip file or in device assembly.

    tools/kernel_regs.py crdmodel_amd/csrc/crd_fused.hip [extra hipcc flags]      # compiles (hipcc -S, device only), prints the table
    tools/kernel_regs.py --asm build/x-hip-amdgcn-amd-amdhsa-gfx950.s ...         # reads assembly the build kept (-save-temps=obj)
        [--table build/crd_kernel_table.inc]   the step kernels' rows as C initialisers (compiled into libcrd: crd_get_launch_geometry)
        [--json profiles/r05/kernel_table.json]

The loop is the kernel's largest loop by vector instructions (the pipeline's steady state: the unrolled iterations of one trip);
counts are static -- instructions in the loop body between its header label and its back edge, whatever branches inside it skip.
VALU = every v_* instruction (DPP moves included); what issue costs a launch is VALU x trips x 4 cycles per wavefront (bench.py:
roofline.issue_frac)."""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_asm(src, flags):
    out = "/tmp/kernel_regs_%d.s" % os.getpid()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "crdmodel_amd", "csrc"), "--offload-device-only", "-S", src, "-o", out] + flags
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(res.stderr[-3000:])
    text = open(out).read()
    os.unlink(out)
    return text


def demangle(names):
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::|crd::|^void ", "", r).split("(")[0] for r in res]


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_sleep")):
        return "salu"
    return "other"


VMEM_OP = re.compile(r"\s*(buffer_|global_|flat_|scratch_)")


def exec_skipped_vmem(body):
    """Vector-memory instructions a wavefront can SKIP on its execution mask: [(line, branch, first skipped instruction)].

    The multi-step pipelines wait for their LDS-DMA row fills with hand-counted `s_waitcnt vmcnt(N)` (crd_fused_impl.h: ring_read,
    kWaitFill / kWaitSteady): N is the number of vector-memory operations the wavefront ISSUES between a slot's fill and its read.
    The count must not depend on the execution mask: a `s_cbranch_execz` that jumps over a store (what the compiler forms around
    `if (lane_stores) store`) leaves a wavefront without storing lanes with fewer operations in flight than N, the wait then waits
    for nothing and the read can overtake the fill.  A region skipped as a whole is harmless when it drains what it issued
    (`s_waitcnt vmcnt(0)` behind its last vector-memory instruction: the whole work item under the per-chunk absorbing-rows
    decision is such a region)."""
    label_at = {}
    for k, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = k
    found = []
    for k, ln in enumerate(body):
        m = re.match(r"\s*s_cbranch_exec(?:z|nz)\s+(\S+)", ln)
        if not m:
            continue
        t = label_at.get(m.group(1))
        if t is None or t <= k:
            continue  # a loop's back edge: nothing is skipped
        vm = [q for q in range(k + 1, t) if VMEM_OP.match(body[q])]
        if not vm:
            continue
        if any(re.match(r"\s*s_waitcnt\s+vmcnt\(0\)", body[q]) for q in range(vm[-1] + 1, t)):
            continue
        found.append((k, ln.strip(), body[vm[0]].strip()))
    return found


def _regs_of(ln):
    """The vector registers an instruction line mentions: {numbers}."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", ln):
        out.update(range(int(a), int(b) + 1))
    out.update(int(n) for n in re.findall(r"\bv(\d+)\b", ln))
    return out


def async_lds_read_hazards(body):
    """LDS reads an asm block issues WITHOUT waiting for them, whose destination registers something touches before an asm block's
    `s_waitcnt lgkmcnt(0)`: [(line, register, offending instruction)].

    The block-strip kernels issue the reads of the neighbours' edge values at the END of an iteration (crd_fused_impl.h: edge_exchange) and
    wait for them at the start of the next (ring_read_with_edges), so that the LDS latency hides under the loop's turn-around.  The
    compiler does not know that the first asm's outputs are not there yet: were it to copy or spill one of those registers between the
    two asm blocks -- at the loop's back edge, say -- the copy would read a register whose data has not landed (round 6 met exactly that
    when it tried the same for the ring reads: results changed from run to run).  Nothing in the language forbids it, so the build
    checks the assembly: from each such read, along fall-through and taken branches, up to the waiting asm block, no instruction may
    mention a destination register."""
    n = len(body)
    label_at = {}
    for k, ln in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = k
    in_asm = [False] * n
    inside = False
    for k, ln in enumerate(body):
        if "#ASMSTART" in ln:
            inside = True
        in_asm[k] = inside
        if "#ASMEND" in ln:
            inside = False
    found = []
    for k, ln in enumerate(body):
        t = ln.strip()
        if not (in_asm[k] and t.startswith("ds_read")):
            continue
        # does this asm block wait for its own reads?
        e = k
        while e < n and "#ASMEND" not in body[e]:
            e += 1
        if any(re.match(r"\s*s_waitcnt\b.*lgkmcnt\(0\)", body[q]) for q in range(k + 1, e)):
            continue
        dst = _regs_of(t.split(",")[0])
        seen, stack = set(), [e + 1]
        while stack:
            q = stack.pop()
            steps = 0
            while q < n and steps < 4000:
                if q in seen:
                    break
                seen.add(q)
                steps += 1
                u = body[q].strip()
                if in_asm[q] and re.match(r"s_waitcnt\b.*lgkmcnt\(0\)", u):
                    break  # landed
                if u and not u.startswith((";", ".")) and not u.endswith(":"):
                    op = u.split()[0]
                    if op == "s_endpgm":
                        break
                    hit = dst & _regs_of(u)
                    # (the reads of one exchange sit in one asm block: its own other reads name other registers)
                    if hit and not (in_asm[q] and op.startswith("ds_read") and not (dst & _regs_of(u.split(",")[0]))):
                        found.append((k, min(hit), u))
                        break
                    mb = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", u)
                    if mb and mb.group(1) in label_at:
                        stack.append(label_at[mb.group(1)])
                        if op == "s_branch":
                            break
                q += 1
    return found


def parse(text):
    """[{name (mangled), vgprs, sgprs, scratch, occupancy, lds, loop: {valu, salu, vmem, lds, other, total}}] for every kernel."""
    kernels = []
    lines = text.splitlines()
    starts = [i for i, ln in enumerate(lines) if re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln) and not ln.startswith(".")]
    for n, i in enumerate(starts):
        name = lines[i].split(":")[0]
        end = starts[n + 1] if n + 1 < len(starts) else len(lines)
        body = lines[i:end]
        meta = {}
        for ln in body:
            m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", ln)
            if m:
                meta[m.group(1)] = int(m.group(2))
        if "NumVgprs" not in meta:
            continue  # not a kernel (a label of some other kind)
        # loops: header labels and the last branch back to each
        label_at = {}
        for k, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                label_at[m.group(1)] = k
        best = None
        for label, k0 in label_at.items():
            if "Loop Header" not in body[k0] and not (k0 + 1 < len(body) and body[k0 + 1].lstrip().startswith(";") and "Loop Header" in body[k0 + 1]):
                continue
            back = [k for k in range(k0, len(body)) if re.match(r"\s*s_c?branch\w*\s+" + re.escape(label) + r"\s*$", body[k])]
            if not back:
                continue
            mix = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0, "moves": 0, "nops": 0}
            for ln in body[k0:back[-1] + 1]:
                ln = ln.strip()
                if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
                    continue
                op = ln.split()[0]
                mix[classify(op)] += 1
                # plain register moves (not the DPP lane shifts): what the three-address stage updates and the vector-register
                # constant keep out of the loop (crd_fused_impl.h: stage_fma; crd_device.h: in_vector_registers) -- tests watch the count
                if op.startswith("v_mov_b") and "dpp" not in ln and "row_" not in ln and "wave_sh" not in ln:
                    mix["moves"] += 1
                if op == "s_nop":
                    mix["nops"] += 1
            mix["total"] = sum(v for k_, v in mix.items() if k_ not in ("moves", "nops"))
            if best is None or mix["valu"] > best["valu"]:
                best = mix
        kernels.append({"mangled": name, "exec_skipped_vmem": len(exec_skipped_vmem(body)), "async_lds_read_hazards": len(async_lds_read_hazards(body)),
                        "vgprs": meta.get("NumVgprs", 0), "sgprs": meta.get("TotalNumSgprs", meta.get("NumSgprs", 0)), "scratch": meta.get("ScratchSize", 0),
                        "occupancy": meta.get("Occupancy", 0), "lds": meta.get("LDSByteSize", 0), "loop": best or {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0, "moves": 0, "nops": 0, "total": 0}})
    for k, nm in zip(kernels, demangle([k["mangled"] for k in kernels])):
        k["name"] = nm
    return kernels


def table_digest(rows):
    """16 hex digits over the step kernels' rows (template arguments, registers, occupancy, loop instruction mix): two builds whose
    kernels the assembler printed alike have the same digest; a changed kernel changes it.  Profile tables (profiles/pmc_traffic.json,
    plan_stats.json) carry the digest of the build they were measured on."""
    import hashlib

    canon = json.dumps(sorted(rows, key=lambda r: (r["precision"], r["model"], r["absorb"], r["embed"], r["cols"], r["nt"], r["steps"])), sort_keys=True)
    return hashlib.sha256(canon.encode()).hexdigest()[:16]


STEP_KERNEL = re.compile(r"crd_rk4_fused_step_kernel<(double|float), (\d+), (true|false), (\d+), (\d+), (true|false), (\d+)>")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", nargs="?")
    ap.add_argument("--asm", nargs="*", default=[])
    ap.add_argument("--table", default="")
    ap.add_argument("--json", default="")
    a, extra = ap.parse_known_args()
    kernels = []
    if a.source:
        kernels += parse(compile_to_asm(a.source, extra))
    for path in a.asm:
        kernels += parse(open(path).read())
    print("%-64s %5s %5s %7s %4s %6s | loop: %5s %5s %5s %4s %6s | %s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS", "VALU", "SALU", "VMEM", "LDS", "total",
                                                                         "exec-skipped VMEM"))
    for k in kernels:
        lp = k["loop"]
        print("%-64s %5d %5d %7d %4d %6d | %11d %5d %5d %4d %6d | %d" % (k["name"][:64], k["vgprs"], k["sgprs"], k["scratch"], k["occupancy"], k["lds"], lp["valu"], lp["salu"],
                                                                          lp["vmem"], lp["lds"], lp["total"], k["exec_skipped_vmem"]) + (" | async LDS read hazards: %d" % k["async_lds_read_hazards"] if k["async_lds_read_hazards"] else ""))
    rows = []
    for k in kernels:
        m = STEP_KERNEL.search(k["name"])
        if m:
            real, model, absorb, embed, cols, nt, steps = m.groups()
            rows.append({"precision": "f64" if real == "double" else "f32", "model": int(model), "absorb": int(absorb == "true"), "embed": int(embed), "cols": int(cols),
                         "nt": int(nt == "true"), "steps": int(steps), "vgprs": k["vgprs"], "sgprs": k["sgprs"], "lds_bytes": k["lds"], "scratch_bytes": k["scratch"],
                         "wavefronts_per_simd": k["occupancy"], "loop": k["loop"], "exec_skipped_vmem": k["exec_skipped_vmem"],
                         "async_lds_read_hazards": k["async_lds_read_hazards"]})
    if a.table:
        with open(a.table, "w") as f:
            f.write("// generated by tools/kernel_regs.py from the assembly of this build's step kernels -- do not edit\n")
            for r in rows:
                lp = r["loop"]
                f.write("{%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d},\n" % (
                    8 if r["precision"] == "f64" else 4, r["model"], r["absorb"], r["embed"], r["cols"], r["nt"], r["steps"], r["vgprs"], r["sgprs"], r["lds_bytes"],
                    r["scratch_bytes"], r["wavefronts_per_simd"], lp["valu"], lp["salu"], lp["vmem"], lp["lds"], lp["total"], r["exec_skipped_vmem"]))
            # the digest of the rows above: what profiles/*.json entries are stamped with (crd_kernel_table_digest, bench.py)
            f.write("#define CRD_KERNEL_TABLE_DIGEST \"%s\"\n" % table_digest(rows))
    if a.json:
        json.dump({"digest": table_digest(rows), "_comment": "step kernels of this build: registers, occupancy and the static instruction mix of the steady-state loop (one trip = the unrolled "
                               "pipeline iterations), from the code object's assembly (tools/kernel_regs.py)", "kernels": rows}, open(a.json, "w"), indent=1)


    # The vmcnt contract of the multi-step pipelines (exec_skipped_vmem above): a build that breaks it does not go on.
    for r in rows:
        if r["async_lds_read_hazards"]:
            sys.stderr.write("kernel_regs.py: %s model %d absorb %d cols %d nt %d steps %d: %d register(s) with LDS reads in flight touched before the asm block that waits "
                             "for them\n" % (r["precision"], r["model"], r["absorb"], r["cols"], r["nt"], r["steps"], r["async_lds_read_hazards"]))
    if any(r["async_lds_read_hazards"] for r in rows):
        sys.exit(4)
    broken = [r for r in rows if r["steps"] >= 2 and r["exec_skipped_vmem"]]
    for r in broken:
        sys.stderr.write("kernel_regs.py: %s model %d absorb %d cols %d nt %d steps %d: %d vector-memory region(s) skipped on the execution mask -- the hand-counted "
                         "s_waitcnt vmcnt of its ring reads no longer holds\n" % (r["precision"], r["model"], r["absorb"], r["cols"], r["nt"], r["steps"], r["exec_skipped_vmem"]))
    if broken:
        sys.exit(3)


if __name__ == "__main__":
    main()
