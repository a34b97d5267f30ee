# Round 6: `rocprofv3 --kernel-trace --stats` of bench.py ITSELF with every launch plan pinned in turn (bench.py --launch-plan): the
# step kernel's row of each run's kernel_stats.csv next to the HIP-event kernel time the same run printed.
# Lands in gpurun_out/r06/plan_stats/{rows.csv, plan_stats_bench.json}.   usage: bash tools/jobs/r06_plan_stats.sh [bench args...]
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06/plan_stats; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 - "$@" <<'PY'
import csv, glob, json, os, re, subprocess, sys, shutil
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R)
import crdmodel_amd as crd
extra = sys.argv[1:]
model = extra[extra.index("--model") + 1] if "--model" in extra else "fhn"
prec = extra[extra.index("--precision") + 1] if "--precision" in extra else "f64"
out_dir = os.path.join(R, "gpurun_out", "r06", "plan_stats")
rows_path = os.path.join(out_dir, "rows_%s_%s.csv" % (model, prec))
recs, header_done = {}, False
with open(rows_path, "w") as rows:
    plans = crd.launch_plan_candidates()
    if os.environ.get("PLANS"):  # "mode,mapping,cols,nt,steps;...": only these
        plans = [tuple(int(v) for v in q.split(",")) for q in os.environ["PLANS"].split(";") if q]
    plans = [q for q in plans if q[4] != 3 or (model == "fhn" and q[2] == (1 if prec == "f64" else 2))]
    for plan in plans:
        key = crd.plan_key(model, prec, plan)
        d = "/tmp/ps_run"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(R, "bench.py"), "--steps", "198", "--warmup", "18",
               "--no-cpu-baseline", "--staged-steps", "0", "--launch-plan", ",".join(str(v) for v in plan)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        line = None
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                line = json.loads(ln)
                break
        stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if r.returncode != 0 or line is None or not stats:
            print("FAILED", key, r.returncode, r.stderr[-500:], flush=True)
            continue
        best = None
        for row in csv.DictReader(open(stats[0])):
            if "crd_rk4_fused_step_kernel" in row["Name"]:
                if not header_done:
                    rows.write("plan_key," + ",".join(row.keys()) + "\n")
                    header_done = True
                rows.write(key + "," + ",".join('"%s"' % v if "," in v else v for v in row.values()) + "\n")
                rows.flush()
                if best is None or int(row["Calls"]) > int(best["Calls"]):  # (a two-step plan also has a one-step row: an odd last step)
                    best = row
        if best is not None:
            row = best
            if True:
                recs[key] = {"bench_stats_avg_us": float(row["AverageNs"]) / 1e3, "bench_stats_min_us": float(row["MinNs"]) / 1e3, "bench_stats_max_us": float(row["MaxNs"]) / 1e3,
                             "bench_stats_calls": int(row["Calls"]), "bench_stats_kernel": re.sub(r"\(anonymous namespace\)::|crd::|void ", "", row["Name"]).split("(")[0], "bench_kernel_ms_events": line["roofline"]["kernel_ms"], "kernel_digest": line["roofline"].get("kernel_digest", ""), "bench_ms_per_step": line["ms_per_step"],
                             "bench_frac": line["roofline"]["frac"], "bench_frac_wall": line["roofline"]["frac_wall"], "bench_value": line["value"],
                             "bench_command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 198 --warmup 18 --no-cpu-baseline --staged-steps 0 --launch-plan %s %s"
                                              % (",".join(str(v) for v in plan), " ".join(extra)),
                             "bench_rows": "profiles/r06/plan_stats/" + os.path.basename(rows_path)}
        print(key, recs.get(key), flush=True)
json.dump(recs, open(os.path.join(out_dir, "plan_stats_bench_%s_%s.json" % (model, prec)), "w"), indent=1)
PY
