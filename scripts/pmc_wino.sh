# HBM traffic of the Winograd batched GEMM (and the grouped weight gradient beside it): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes over scripts/pmc_conv_group.py (MI355X_MICROARCH.md) -> gpurun_out/r03_pmc_wino_gemm_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace --output-format csv"
$R --pmc FETCH_SIZE -d gpurun_out/pmcw/fetch -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 || exit 1
$R --pmc WRITE_SIZE -d gpurun_out/pmcw/write -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 || exit 1
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmcw/sq -o p -- python3 scripts/pmc_conv_group.py > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
def P(match, *d):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_parse.py")] +
                                              [os.path.join(ROOT, "gpurun_out", "pmcw", x) for x in d] + ["--match", match]))
T, C = 4 * (64 * 64 + 32 * 32 + 16 * 16 + 8 * 8 + 4 * 4), 256
# algorithmic bytes: GEMM reads V (T,C) and U (C,C), writes M (T,C) per position; the weight gradient reads dM and V, writes dU
ALG = {"k_gemm_batched": 16 * (T * C * 4 * 2 + C * C * 4), "k_wgrad_batched": 16 * (T * C * 4 * 2 + C * C * 4)}
out = {}
for match, alg in ALG.items():
    f, w, sq = P(match, "fetch"), P(match, "write"), P(match, "sq")
    if not f:
        continue
    k = list(f)[0]
    rd, wr = f[k]["FETCH_SIZE"] * 1024 * 2, w[k]["WRITE_SIZE"] * 1024
    cyc = sq[k]["SQ_BUSY_CYCLES"] / 32.0
    out[k] = {"FETCH_SIZE_KB_raw": f[k]["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w[k]["WRITE_SIZE"], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
              "traffic_bytes": rd + wr, "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr) / alg,
              "mfma_utilisation": sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), "clock_GHz": cyc / sq[k]["duration_ns_under_pmc"],
              "duration_us_under_pmc": sq[k]["duration_ns_under_pmc"] / 1e3}
    print(k, json.dumps(out[k]))
d = {"shape": f"16 GEMMs ({T} x 256) @ (256 x 256), float32: the Winograd products of the RPN head conv on 4 x 512 x 512 images (k_gemm_batched_f32: "
              "forward / backward-data; k_wgrad_batched_f32: dU[k] = dM[k]^T V[k], f32 atomics of 12 pixel splits into (256 x 256) per position)",
     "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes over scripts/pmc_conv_group.py; FETCH_SIZE x 2 (gfx950)",
     "kernels": out}
json.dump(d, open(os.path.join(ROOT, "gpurun_out", "r03_pmc_wino_gemm_traffic.json"), "w"), indent=1)
PY
rm -rf gpurun_out/pmcw
