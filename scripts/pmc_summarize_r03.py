"""gpurun_out/pmc3/* (the passes of scripts/pmc_r03.sh) -> profiles/r03_pmc_conv_traffic_fp32.json, r03_pmc_geometry_argmax_traffic.json
HBM-side traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE from separate passes, in KB;
FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), WRITE_SIZE as reported.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *d, m="k_conv": json.loads(subprocess.check_output(
    [sys.executable, os.path.join(ROOT, "scripts", "pmc_parse.py")] + [os.path.join(ROOT, "gpurun_out", "pmc3", x) for x in d] + ["--match", m]))
PYRAMID = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
px = 4 * sum(h * w for h, w in PYRAMID) * 256
alg = px * 4 * 2 + 2.36e6                    # x (or dy) read + y (or dx) written / x and dy read; weights or dW 2.36 MB
f, w, sq = P("grp_fetch"), P("grp_write"), P("grp_sq")
out = {"shape": "grouped 3x3 conv 256->256 over the five pyramid levels of 4 x 512 x 512 images, f32 (activations %.1f MB per tensor)" % (px * 4 / 1e6),
       "method": __doc__.split("\n", 1)[1].strip(), "kernels": {}}
for k in sq:
    if "grp" not in k:
        continue
    rd, wr = f[k]["FETCH_SIZE"] * 1024 * 2, w[k]["WRITE_SIZE"] * 1024
    cyc = sq[k]["SQ_BUSY_CYCLES"] / 32.0
    out["kernels"][k] = {
        "FETCH_SIZE_KB_raw": f[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w[k]["WRITE_SIZE"], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
        "traffic_bytes": rd + wr, "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "SQ_VALU_MFMA_BUSY_CYCLES": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": sq[k]["SQ_BUSY_CYCLES"],
        "SQ_INSTS_VALU_MFMA_MOPS_F32": sq[k].get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "kernel_cycles": cyc,
        "clock_GHz": cyc / sq[k]["duration_ns_under_pmc"], "mfma_utilisation": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
        "wave_cycles_parked_frac": sq[k]["SQ_WAIT_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
        "wave_cycles_issue_stall_frac": sq[k]["SQ_WAIT_INST_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
        "wave_cycles_issuing_frac": sq[k]["SQ_ACTIVE_INST_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
        "duration_us_under_pmc": sq[k]["duration_ns_under_pmc"] / 1e3}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_pmc_conv_traffic_fp32.json"), "w"), indent=1)
g = P("geoa_fetch", "geoa_write", m="k_project")
k = list(g)[0]
rd, wr = g[k]["FETCH_SIZE"] * 1024 * 2, g[k]["WRITE_SIZE"] * 1024
galg = 1024 * 1000 * 60 + 1024 * (16 + 12 + 12 + 32 + 8 + 4)
geo = {"shape": "k_project_score<4>, argmax-only outputs: 1024 objects x 1000 cubes (60 B read per cube, 12 B written per object)",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --workload geometry --argmax-only`; FETCH_SIZE x 2 (gfx950)",
       "kernels": {k: {"FETCH_SIZE_KB_raw": g[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": g[k]["WRITE_SIZE"], "hbm_read_bytes": rd,
                       "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": galg,
                       "traffic_over_algorithmic": (rd + wr) / galg, "duration_us_under_pmc": g[k]["duration_ns_under_pmc"] / 1e3}}}
json.dump(geo, open(os.path.join(ROOT, "profiles", "r03_pmc_geometry_argmax_traffic.json"), "w"), indent=1)
for name, d in (("fp32 grouped", out), ("geometry argmax-only", geo)):
    for k, v in d["kernels"].items():
        print(name, k, "traffic %.1f MB (x%.2f algorithmic)" % (v["traffic_bytes"] / 1e6, v["traffic_over_algorithmic"]),
              ("MFMA util %.3f clock %.2f GHz" % (v["mfma_utilisation"], v["clock_GHz"])) if "mfma_utilisation" in v else "")
