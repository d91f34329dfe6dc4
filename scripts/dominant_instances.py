"""per-instance rocprofv3 durations of the dominant layer's kernels inside the bench's instrumented train steps, next to the
HIP-event figures bench.py printed in the SAME run (profiles/r02_dominant_kernel_in_step.json).
    python scripts/dominant_instances.py RESULTS.db BENCH_STDOUT.json OUT.json
The dominant layer (3x3 256->256 on 4x128x128) is launched twice per step and direction (FPN p2 output conv, RPN head on p2);
among the launches of a kernel in the last three steps of a phase its instances are the ones within 20 % of the longest."""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = db.execute("select name, start, end, grid_x / workgroup_x from kernels order by start").fetchall()
cut = next((i for i, r in enumerate(rows) if "unsigned short" in r[0]), len(rows))
out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline", "phases": {}}
for phase, part, roof in (("fp32", rows[:cut], bench["roofline"]), ("bf16", rows[cut:], bench.get("bf16_mode", {}).get("roofline", {}))):
    ends = [i for i, r in enumerate(part) if "k_sgd" in r[0]]
    if len(ends) < 8:                                    # phase not run (CR_BENCH_BF16=0)
        continue
    bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] != i + 1]
    last3 = part[bounds[-4] + 1:bounds[-1] + 1]
    ph = {}
    for key in ("k_conv_wgrad", "k_conv_igemm_dma<128, 3, 0", "k_conv_igemm_dma<128, 3, 1"):
        d = [(r[2] - r[1]) / 1e3 for r in last3 if key in r[0] and ("wgrad" not in key or "<128, 3" in r[0])]
        if not d:
            continue
        big = [x for x in d if x >= 0.8 * max(d)]
        ph[key] = {"instances_of_the_dominant_layer": len(big), "rocprof_avg_us": sum(big) / len(big), "rocprof_min_us": min(big),
                   "rocprof_max_us": max(big), "all_launches_of_this_kernel_in_3_steps": len(d)}
    ev = roof.get("in_step", {})
    for k, v in ev.items():
        for key in ph:
            if key.replace(" ", "").split("<")[0] in k.replace(" ", "") and (("wgrad" in key) == ("wgrad" in k)) and \
               (("wgrad" in key) or (",0" in k.replace(" ", "")) == key.endswith(" 0")):
                ph[key]["bench_hip_events_avg_us"] = v["ms"] * 1e3
                ph[key]["bench_kernel_name"] = k
    out["phases"][phase] = ph
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
