"""reads a rocprofv3 kernel trace CSV of bench.py and reports, for the last few steps, busy time vs wall time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step boundaries: k_sgd launches (2 per step: two param groups) -> use every 2nd
sgd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sgd")]
ends = sgd[1::2]
for a, b in list(zip(ends[:-1], ends[1:]))[-4:]:
    seg = rows[a + 1:b + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
    big = sorted(gaps, reverse=True)[:5]
    pos = sum(g for g in gaps if g > 0)
    print(f"step: {len(seg)} kernels, wall {(t1-t0)/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {pos/1e6:.3f} ms, "
          f"median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, 5 largest gaps {[round(g/1e3,1) for g in big]} us")
