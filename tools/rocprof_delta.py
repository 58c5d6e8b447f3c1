"""Per-step kernel times from two rocprofv3 runs of bench.py that differ only in --steps:
   python tools/rocprof_delta.py small.db big.db <delta_steps> [out.csv]
Everything that is setup (bank build, roofline/cpu-baseline legs, warmup) cancels; what is left is one step."""
import csv, sqlite3, sys


def table(db_path):
    con = sqlite3.connect(db_path)
    tabs = [r[0] for r in con.execute("SELECT name FROM sqlite_master WHERE type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    out = {}
    for name, n, tot in con.execute(
            f"SELECT s.kernel_name, COUNT(*), SUM(d.end - d.start) FROM {disp} d JOIN {sym} s ON d.kernel_id = s.id GROUP BY s.kernel_name"):
        out[name] = (n, tot)
    return out


def main():
    a, b, steps = table(sys.argv[1]), table(sys.argv[2]), float(sys.argv[3])
    rows = []
    for name in b:
        n0, t0 = a.get(name, (0, 0))
        n1, t1 = b[name]
        if n1 - n0 > 0:
            rows.append((name, (n1 - n0) / steps, (t1 - t0) / steps / 1e3, (t1 - t0) / max(1, n1 - n0) / 1e3))
    rows.sort(key=lambda r: -r[2])
    total = sum(r[2] for r in rows)
    print(f"kernel time per step: {total / 1e3:.3f} ms")
    for name, calls, us, avg in rows:
        print(f"{name[:96]:96s} {calls:7.1f} calls {us:9.1f} us/step {avg:9.1f} us avg {100 * us / total:5.1f}%")
    if len(sys.argv) > 4:
        with open(sys.argv[4], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsPerStep", "UsPerStep", "AverageUs", "Percentage"])
            for name, calls, us, avg in rows:
                w.writerow([name, f"{calls:.2f}", f"{us:.1f}", f"{avg:.1f}", f"{100 * us / total:.2f}"])


if __name__ == "__main__":
    main()
