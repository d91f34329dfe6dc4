"""rocprofv3 --pmc ... --output-format csv  ->  per-kernel medians of every counter (and of the dispatch duration), JSON.
    python scripts/pmc_parse.py DIR_OR_CSV [more ...] [--match k_conv] > out.json"""
import csv, glob, json, os, statistics, sys
csv.field_size_limit(1 << 30)
match = "k_"
args = []
it = iter(sys.argv[1:])
for a in it:
    if a == "--match":
        match = next(it)
    else:
        args.append(a)
out = {}
for a in args:
    files = glob.glob(os.path.join(a, "*counter_collection.csv")) if os.path.isdir(a) else [a]
    for f in files:
        per = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if match not in name or len(name) > 400:
                continue
            short = name.replace("void ", "").split("(")[0]
            per.setdefault(short, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            per[short].setdefault("_ns", []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, cs in per.items():
            d = out.setdefault(k, {})
            for c, v in cs.items():
                d[c if c != "_ns" else "duration_ns_under_pmc"] = statistics.median(v)
                d.setdefault("launches", len(v))
print(json.dumps(out, indent=1))
