"""gpurun_out/pmcg/* (scripts/pmc_geo_fast.sh) -> profiles/r03_pmc_geometry_fast_traffic.json, r03_pmc_geometry_fast_argmax_traffic.json,
r03_geometry_kernel_durations.json.  FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B),
WRITE_SIZE as reported, both in KB (MI355X_MICROARCH.md)."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *d: json.loads(subprocess.check_output(
    [sys.executable, os.path.join(ROOT, "scripts", "pmc_parse.py")] + [os.path.join(ROOT, "gpurun_out", "pmcg", x) for x in d] + ["--match", "k_project"]))
per_obj = 16 + 12 + 12 + 32 + 8 + 4
for v, fname, out_b in (("all", "r03_pmc_geometry_fast_traffic.json", 96), ("none", "r03_pmc_geometry_fast_argmax_traffic.json", 0)):
    g = P(v + "_fetch", v + "_write")
    k = list(g)[0]
    rd, wr = g[k]["FETCH_SIZE"] * 1024 * 2, g[k]["WRITE_SIZE"] * 1024
    alg = 1024 * 1000 * (60 + out_b) + 1024 * per_obj
    d = {"shape": f"{k}: 1024 objects x 1000 cubes, 60 B read + {out_b} B written per cube",
         "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over scripts/geo_one.py fast " + v + "; FETCH_SIZE x 2 (gfx950)",
         "kernels": {k: {"FETCH_SIZE_KB_raw": g[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": g[k]["WRITE_SIZE"], "hbm_read_bytes": rd,
                         "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": alg,
                         "traffic_over_algorithmic": (rd + wr) / alg, "duration_us_under_pmc": g[k]["duration_ns_under_pmc"] / 1e3}}}
    json.dump(d, open(os.path.join(ROOT, "profiles", fname), "w"), indent=1)
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", fname), "w"), indent=1)       # (profiles/ does not travel back from the box)
    print(v, k, "traffic %.1f MB (x%.3f algorithmic)" % ((rd + wr) / 1e6, (rd + wr) / alg))
dur = {}
for n in ("fast_all", "fast_none", "exact_all", "exact_none"):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", "pmcg", "dur_" + n, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_project_score" in r["Kernel_Name"]][20:]
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    dur[n] = {"kernel": rows[0]["Kernel_Name"], "launches": len(t), "mean_us": sum(t) / len(t), "min_us": min(t), "max_us": max(t)}
    print(n, "%.1f us mean, %.1f min" % (dur[n]["mean_us"], dur[n]["min_us"]))
json.dump({"method": "rocprofv3 --kernel-trace over scripts/geo_one.py (220 launches, first 20 dropped), no counters", "variants": dur},
          open(os.path.join(ROOT, "gpurun_out", "r03_geometry_kernel_durations.json"), "w"), indent=1)
