#!/usr/bin/env python3
"""development helper: what tools/profile_configs.sh left under gpurun_out/prof_configs/ -> profiles/r<NN>_kernel_stats_<cfg>.csv,
profiles/r<NN>_pmc_<cfg>.json (counters + derived ratios + the bench line of the profiled run) and one line per config on stdout
usage: python tools/profile_collect_configs.py <round>"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1])
src = os.path.join(ROOT, "gpurun_out", "prof_configs")
dst = os.path.join(ROOT, "profiles")
tag = "r%02d_" % rnd
for cfg in ("c1", "c3a", "c3b", "c4"):
    if not os.path.exists(os.path.join(src, "bench_%s.json" % cfg)):
        continue
    shutil.copy(os.path.join(src, "kernel_stats_%s.csv" % cfg), os.path.join(dst, tag + "kernel_stats_%s.csv" % cfg))
    line = json.load(open(os.path.join(src, "bench_%s.json" % cfg)))
    c = {}
    for i in (1, 2, 3):
        for k, v in json.load(open(os.path.join(src, "pmc_%s_%d.json" % (cfg, i)))).items():
            c[k] = v["sum"]
    seg128 = float(re.search(r'"segments_per_step": ([0-9.]+)', open(os.path.join(src, "pmc_%s_1.log" % cfg)).read()).group(1))
    d = {"wave_segments_of_the_counter_run": seg128 / 64.0,
         "valu_per_wave_segment": c["SQ_INSTS_VALU"] / (seg128 / 64.0), "salu_per_wave_segment": c["SQ_INSTS_SALU"] / (seg128 / 64.0),
         "lane_occupancy = SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU)": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
         "wait_fraction = SQ_WAIT_ANY/SQ_WAVE_CYCLES": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
         "l1_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
         "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
         "scratch_and_state_stores_per_wave_segment (SQ_INSTS_VMEM_WR)": c["SQ_INSTS_VMEM_WR"] / (seg128 / 64.0)}
    # the dominant kernel of the stats file
    with open(os.path.join(src, "kernel_stats_%s.csv" % cfg), newline="") as f:
        rows = [r for r in csv.DictReader(f) if "render_kernel" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    k = rows[0]
    rl = line["roofline"]
    # per launch: the stats file also holds the launches of the warm-up steps, the bench line's counts are those of the timed ones
    launches_per_step = rl["launches"] / float(line["steps"])
    frac = 240.0 * line["config"]["segments_per_step"] / (launches_per_step * float(k["AverageNs"]) * 1e-9 / rl["concurrent_launches"]) / 8e12
    out = {"config": cfg, "workload": line["config"]["workload"], "bench_line_of_the_profiled_run": line, "counters_at_128spp": c, "derived": d,
           "dominant_kernel": {"name": k["Name"][:90], "calls": int(k["Calls"]), "average_ns": float(k["AverageNs"]), "total_ns": float(k["TotalDurationNs"]), "percent": float(k["Percentage"])},
           "frac_recomputed = 240 B x segments per step / (launches per step x average launch time / concurrent launches) / 8 TB/s": frac}
    with open(os.path.join(dst, tag + "pmc_%s.json" % cfg), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("%s: %s | %d launches x %.2f ms, %d in flight | %.3f G segments/s, frac %.4f (bench) %.4f (from the stats file) | lane occupancy %.3f, wait %.3f, L2 hit %.3f, VALU / wave-segment %.0f" % (
        cfg, line["config"]["workload"], int(k["Calls"]), float(k["AverageNs"]) * 1e-6, rl["concurrent_launches"], rl["gsegments_per_s"], rl["frac"], frac,
        d["lane_occupancy = SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU)"], d["wait_fraction = SQ_WAIT_ANY/SQ_WAVE_CYCLES"], d["l2_hit_rate"], d["valu_per_wave_segment"]))
