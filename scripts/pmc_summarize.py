"""gpurun_out/pmc/* (the passes of scripts/pmc_passes.sh) -> profiles/r02_pmc_*.json
HBM-side traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE from separate passes, in KB;
FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), WRITE_SIZE as reported (exact for
16-B stores and float atomics).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles =
SQ_BUSY_CYCLES / 32 shader engines (the quotient reproduces duration x ~2.07 GHz on these kernels)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *d, m="k_conv": json.loads(subprocess.check_output(
    [sys.executable, os.path.join(ROOT, "scripts", "pmc_parse.py")] + [os.path.join(ROOT, "gpurun_out", "pmc", x) for x in d] + ["--match", m]))


def conv(prec, alg):
    f, w, sq = P(prec + "_fetch"), P(prec + "_write"), P(prec + "_sq")
    out = {"shape": alg["shape"], "method": __doc__.split("\n", 1)[1].strip(), "kernels": {}}
    for k in sq:
        rd, wr = f[k]["FETCH_SIZE"] * 1024 * 2, w[k]["WRITE_SIZE"] * 1024
        cyc = sq[k]["SQ_BUSY_CYCLES"] / 32.0
        mops = [v for c, v in sq[k].items() if c.startswith("SQ_INSTS_VALU_MFMA_MOPS")][0]
        out["kernels"][k] = {
            "FETCH_SIZE_KB_raw": f[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w[k]["WRITE_SIZE"], "hbm_read_bytes": rd,
            "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": alg["bytes"][("wgrad" in k)],
            "traffic_over_algorithmic": (rd + wr) / alg["bytes"][("wgrad" in k)],
            "SQ_VALU_MFMA_BUSY_CYCLES": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": sq[k]["SQ_BUSY_CYCLES"],
            "SQ_INSTS_VALU_MFMA_MOPS": mops, "kernel_cycles": cyc, "clock_GHz": cyc / sq[k]["duration_ns_under_pmc"],
            "mfma_utilisation": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
            "wave_cycles_parked_frac": sq[k]["SQ_WAIT_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
            "wave_cycles_issue_stall_frac": sq[k]["SQ_WAIT_INST_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
            "wave_cycles_issuing_frac": sq[k]["SQ_ACTIVE_INST_ANY"] / sq[k]["SQ_WAVE_CYCLES"],
            "duration_us_under_pmc": sq[k]["duration_ns_under_pmc"] / 1e3}
    return out


px = 4 * 128 * 128 * 256
f32 = conv("f32", {"shape": "3x3 conv 256->256 on 4x128x128 f32 (x 67.1 MB, dy / y 67.1 MB, w 2.36 MB, dW 2.36 MB)",
                   "bytes": {False: px * 4 * 2 + 2.36e6, True: px * 4 * 2 + 2.36e6}})
b16 = conv("bf16", {"shape": "3x3 conv 256->256 on 4x128x128 bf16 (x 33.55 MB, dy / y 33.55 MB, w 1.18 MB, dW 2.36 MB f32)",
                    "bytes": {False: px * 2 * 2 + 1.18e6, True: px * 2 * 2 + 2.36e6}})
x3 = conv("x3", {"shape": "3x3 conv 256->256 on 4x128x128, fp32x3 mode: f32 tensors, split-mode kernels (x 67.1 MB, dy / y 67.1 MB, "
                          "w 2.36 MB f32 = 3.54 MB as three bf16 planes, dW 2.36 MB)",
                 "bytes": {False: px * 4 * 2 + 3.54e6, True: px * 4 * 2 + 2.36e6}})
json.dump(x3, open(os.path.join(ROOT, "profiles", "r02_pmc_conv_traffic_fp32x3.json"), "w"), indent=1)
json.dump(f32, open(os.path.join(ROOT, "profiles", "r02_pmc_conv_traffic_fp32.json"), "w"), indent=1)
json.dump(b16, open(os.path.join(ROOT, "profiles", "r02_pmc_conv_traffic_bf16.json"), "w"), indent=1)
g = P("geo_fetch", "geo_write", m="k_project")
k = list(g)[0]
rd, wr = g[k]["FETCH_SIZE"] * 1024 * 2, g[k]["WRITE_SIZE"] * 1024
alg = 1024 * 1000 * 156 + 1024 * (16 + 12 + 12 + 32 + 8 + 4)
geo = {"shape": "k_project_score<4>: 1024 objects x 1000 cubes, full outputs (60 B read + 96 B written per cube)",
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --workload geometry`; FETCH_SIZE x 2 (gfx950)",
       "kernels": {k: {"FETCH_SIZE_KB_raw": g[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": g[k]["WRITE_SIZE"], "hbm_read_bytes": rd,
                       "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": alg,
                       "traffic_over_algorithmic": (rd + wr) / alg, "duration_us_under_pmc": g[k]["duration_ns_under_pmc"] / 1e3}}}
json.dump(geo, open(os.path.join(ROOT, "profiles", "r02_pmc_geometry_traffic.json"), "w"), indent=1)
for name, d in (("fp32", f32), ("fp32x3", x3), ("bf16", b16), ("geometry", geo)):
    for k, v in d["kernels"].items():
        print(name, k, "traffic %.1f MB (x%.2f algorithmic)" % (v["traffic_bytes"] / 1e6, v["traffic_over_algorithmic"]),
              ("MFMA util %.3f clock %.2f GHz" % (v["mfma_utilisation"], v["clock_GHz"])) if "mfma_utilisation" in v else "")
