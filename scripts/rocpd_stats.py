"""kernel statistics of a rocprofv3 run stored in its default rocpd (SQLite) format -> the per-kernel summary that
`rocprofv3 --stats --output-format csv` writes (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs), as CSV.

    python scripts/rocpd_stats.py RESULTS.db [OUT.csv] [--last-steps N --marker k_sgd] [--by-grid KERNEL_SUBSTRING]
                                  [--until-first SUBSTRING] [--from-first SUBSTRING] [--skip-last N]

With --last-steps the statistics cover only the dispatches of the last N train steps (a step ends with the last `marker`
kernel of the optimizer update), i.e. the timed region without warm-up, capture and the roofline micro-benchmark."""
import csv
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {}
    it = iter(sys.argv[1:])
    for a in it:
        if a.startswith("--"):
            opts[a[2:]] = next(it)
    db = sqlite3.connect(args[0])
    by_grid = "by-grid" in opts            # one row per (kernel, grid size): the instances of one kernel on different layers
    rows = db.execute("select name, start, end, grid_x / workgroup_x from kernels order by start").fetchall()
    if "until-first" in opts:               # e.g. "unsigned short": cut the run at the first bf16 kernel (the fp32 phase of bench.py)
        cut = next((i for i, r in enumerate(rows) if opts["until-first"] in r[0]), len(rows))
        rows = rows[:cut]
    if "from-first" in opts:
        cut = next((i for i, r in enumerate(rows) if opts["from-first"] in r[0]), 0)
        rows = rows[cut:]
    marker = opts.get("marker", "k_sgd")
    if "last-steps" in opts:
        n = int(opts["last-steps"])
        ends = [i for i, r in enumerate(rows) if marker in r[0]]
        # the optimizer update is several launches of the marker kernel in a row: a step boundary is the last of a run
        bounds = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] != i + 1]
        skip = int(opts.get("skip-last", 0))
        if skip:
            bounds = bounds[:-skip]
        lo, hi = bounds[-n - 1] + 1, bounds[-1] + 1
        rows = rows[lo:hi]
        wall = rows[-1][2] - rows[0][1]
        print(f"# {n} steps: {len(rows)} dispatches ({len(rows) / n:.0f} per step), wall {wall / n / 1e6:.3f} ms per step, "
              f"kernel time {sum(r[2] - r[1] for r in rows) / n / 1e6:.3f} ms per step", file=sys.stderr)
    agg = {}
    for name, s, e, blocks in rows:
        if by_grid:
            if opts["by-grid"] not in name:
                continue
            name = f"{name} [grid {blocks} blocks]"
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    out = open(args[1], "w", newline="") if len(args) > 1 else sys.stdout
    w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, a[0], a[1], round(a[1] / a[0], 3), round(100.0 * a[1] / tot, 2), a[2], a[3]])


if __name__ == "__main__":
    main()
