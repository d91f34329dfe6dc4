"""rocprofv3 kernel trace CSV of a bench.py run -> for the last full step (between k_sgd launches): busy / idle time and the
largest idle gaps with the kernels on either side.  python scripts/gap_where.py trace.csv [n_gaps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
sgd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sgd")]
per = 2 if len(sgd) > 1 and sgd[1] - sgd[0] < 5 else 1
ends = sgd[per - 1::per]
a, b = ends[-3], ends[-2]
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]), i) for i in range(len(seg) - 1)]
print(f"{len(seg)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {sum(g for g, _ in gaps if g > 0) / 1e6:.3f} ms")
small = sum(g for g, _ in gaps if 0 < g <= 20000)
print(f"idle in gaps <= 20 us: {small / 1e6:.3f} ms over {sum(1 for g, _ in gaps if 0 < g <= 20000)} gaps")
for g, i in sorted(gaps, reverse=True)[:n]:
    print(f"{g / 1e3:9.1f} us at +{(int(seg[i]['End_Timestamp']) - t0) / 1e6:7.3f} ms  after {seg[i]['Kernel_Name'][:48]:48s} before {seg[i + 1]['Kernel_Name'][:48]}")
