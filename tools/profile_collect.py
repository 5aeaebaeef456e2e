#!/usr/bin/env python3
"""development helper: turns what tools/profile_round.sh left under gpurun_out/prof_round/ into the files committed under
profiles/ (r<NN>_kernel_stats.csv, _bench_under_rocprof.json, _traffic.json, _pmc_render_kernel.json, _pmc_render_kernel_dragon.json).
usage: python tools/profile_collect.py <round number> [source dir = gpurun_out/prof_round]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1])
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof_round")
dst = os.path.join(ROOT, "profiles")
tag = "r%02d_" % rnd


def load(name):
    with open(os.path.join(src, name)) as f:
        return json.load(f)


def sums(names):
    out = {}
    for n in names:
        for k, v in load(n).items():
            out[k] = v["sum"]
    return out


def segments_of(log):
    """segments_per_step of the JSON line a bench.py run left in its log"""
    m = re.search(r'"segments_per_step": ([0-9.]+)', open(os.path.join(src, log)).read())
    return float(m.group(1))


def derived(c, segments=None):
    d = {}
    if segments:
        d["wave_segments"] = segments / 64.0
        d["valu_per_wave_segment"] = c["SQ_INSTS_VALU"] / d["wave_segments"]
        d["salu_per_wave_segment"] = c["SQ_INSTS_SALU"] / d["wave_segments"]
    d["lane_occupancy = SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU)"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    d["wait_fraction = SQ_WAIT_ANY/SQ_WAVE_CYCLES"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    d["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQC_ICACHE_REQ" in c:
        d["icache_miss_rate"] = c["SQC_ICACHE_MISSES"] / c["SQC_ICACHE_REQ"]
    return d


shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, tag + "kernel_stats.csv"))
line = load("bench_under_rocprof.json")
with open(os.path.join(dst, tag + "bench_under_rocprof.json"), "w") as f:
    json.dump(line, f)
    f.write("\n")

fetch, write = load("pmc_FETCH_SIZE.json")["FETCH_SIZE"], load("pmc_WRITE_SIZE.json")["WRITE_SIZE"]
seg_per_step = line["config"]["segments_per_step"]
launches_per_step = line["roofline"]["launches"] / line["steps"]
traffic = {
    "workload": "1920x1080_1024spp_cornell_diffuse.json",
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0   (tools/profile_round.sh)",
    "kernel": "prt::render_kernel<3u,false,6> (lane machine, deferred leaves)",
    "dispatches": fetch["dispatches"],
    "FETCH_SIZE_bytes_per_launch_raw": round(fetch["sum"] * 1024.0 / fetch["dispatches"]),
    "WRITE_SIZE_bytes_per_launch": round(write["sum"] * 1024.0 / write["dispatches"]),
    "note": "one launch = up to 512 frames of ONE of the two interleaved tile sets (half of the frame's pixels; two such launches "
            "are in flight in a normal run, the counter passes serialise them). Counters are in KiB; FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads; an upper bound for the gather part); "
            "WRITE_SIZE as reported. The state stream itself is 83 MB in + 100 MB out per launch; the rest of the reads is BVH "
            "gathers that miss L2, the rest of the writes is scratch (register spill) write-back: see scratch_* below.",
    "algorithmic_bytes_per_launch": round(240.0 * seg_per_step / launches_per_step),
}
traffic["hbm_bytes_per_launch"] = 2 * traffic["FETCH_SIZE_bytes_per_launch_raw"] + traffic["WRITE_SIZE_bytes_per_launch"]
# where the writes come from: vector-memory STORE instructions of the kernel are the 6 state / image stores of a lane at the end of a
# launch plus the scratch stores of its spills (the 128 spp counter pass; 256 B per wave-level dword store)
try:
    c3 = load("pmc_set3.json")
    wr, disp = c3["SQ_INSTS_VMEM_WR"]["sum"], c3["SQ_INSTS_VMEM_WR"]["dispatches"]
    state_stores = 6.0 * (1920 * 1080 / 64.0) / 2.0                      # per launch of one of the two tile sets
    traffic["store_instructions_per_launch"] = round(wr / disp)
    traffic["state_store_instructions_per_launch"] = round(state_stores)
    traffic["scratch_store_bytes_per_launch_estimate"] = round((wr / disp - state_stores) * 256.0)
except Exception:
    pass
with open(os.path.join(dst, tag + "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)

c = sums(["pmc_set%d.json" % k for k in range(1, 6)])
cornell = {
    "workload": "cornell_diffuse.json 1920x1080, 128 spp (counter passes serialise the two streams: one launch at a time)",
    "workload_key": "1920x1080_cornell_diffuse.json", "spp": 128,
    "kernel": "render_kernel<LIGHT|DIFF, 6 waves> (lane machine with deferred leaves, walk_min_lanes 8 / shadow phases in lock step, tri_q 4, 512 frames per launch)",
    "counters": c,
    "derived": derived(c, segments_of("pmc_set1.log")),
}
with open(os.path.join(dst, tag + "pmc_render_kernel.json"), "w") as f:
    json.dump(cornell, f, indent=1)

c = sums(["pmc_dragon%d.json" % k for k in range(1, 4)])
dragon = {
    "workload": "cornell_dragon.json (871 k-triangle stand-in) 3840x2160, 8 spp",
    "workload_key": "3840x2160_cornell_dragon.json", "spp": 8,
    "kernel": "render_kernel<LIGHT|DIFF, 6 waves> (lane machine with deferred leaves, walk_min_lanes 20 / shadow_min_lanes 12, tri_q 4, 4096 frames per launch)",
    "counters": c,
    "derived": derived(c, segments_of("pmc_dragon1.log")),
}
with open(os.path.join(dst, tag + "pmc_render_kernel_dragon.json"), "w") as f:
    json.dump(dragon, f, indent=1)
print(json.dumps({"cornell": cornell["derived"], "dragon": dragon["derived"], "traffic": traffic["hbm_bytes_per_launch"]}, indent=1))
