#!/usr/bin/env python3
"""profiles/valu_busy.json from the SQ counter passes of tools/jobs/r05_sq.sh (profiles/r05/sq/<config>_SQ_{INSTS_VALU,WAVE_CYCLES}.json):
per instantiation (model / precision / columns per lane / steps per launch) how busy the vector ALUs are over a launch and how a
wavefront's cycles split into issuing, stalled at issue and parked at a wait.

    tools/sq_summary.py profiles/r05/sq"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = {"_comment": "How busy the vector ALUs are during a launch of the step kernel: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs) over the launch's cycles (GRBM_GUI_ACTIVE / 8 XCDs), "
                   "from rocprofv3 --pmc passes of tools/plan_sweep.py with the plan pinned (tools/jobs/r05_sq.sh; per-plan records under " + os.path.relpath(src, ROOT) + "/).  "
                   "Keyed by model/precision/columns-per-lane/steps-per-launch; bench.py quotes the entry of the kernel its run used beside its own issue_frac (static "
                   "instruction count x launch geometry): the counter is in units of four cycles and overlapping issue of consecutive wave-instructions is counted twice, so it "
                   "reads high (up to 1.1); wave_cycles_*: a wavefront's cycles issuing / stalled at issue / parked at s_waitcnt or a barrier."}
for a_path in sorted(glob.glob(os.path.join(src, "*_SQ_INSTS_VALU.json"))):
    b_path = a_path.replace("_SQ_INSTS_VALU", "_SQ_WAVE_CYCLES")
    a, b = json.load(open(a_path)), json.load(open(b_path))
    model, prec, size = re.match(r"(\w+?)_(f\d\d)_(\d+)_SQ", os.path.basename(a_path)).groups()
    for key, ra in a["plans"].items():
        ca, cb = ra["counters"], b["plans"][key]["counters"]
        pts = int(size) ** 2
        spl = ra["steps_per_launch"]
        name = "%s/%s/cols%d/steps%d" % (model, prec, ra["plan"][2], spl)
        rec = {"valu_busy": round(ca["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (ca["GRBM_GUI_ACTIVE"] / 8), 3),
               "valu_wave_instructions_per_grid_point_step": round(ca["SQ_INSTS_VALU"] / pts / spl, 3),
               "wave_cycles_issuing": round(cb["SQ_ACTIVE_INST_ANY"] / cb["SQ_WAVE_CYCLES"], 3), "wave_cycles_issue_stalled": round(cb["SQ_WAIT_INST_ANY"] / cb["SQ_WAVE_CYCLES"], 3),
               "wave_cycles_parked": round(cb["SQ_WAIT_ANY"] / cb["SQ_WAVE_CYCLES"], 3), "plan": key, "grid": a["grid"] if "grid" in a else ra.get("grid"),
               "source": os.path.relpath(a_path, ROOT).replace("_SQ_INSTS_VALU", "_*")}
        if name not in out or rec["valu_busy"] > out[name]["valu_busy"]:
            out[name] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_busy.json"), "w"), indent=1)
for k, v in out.items():
    if k != "_comment":
        print(k, v)
