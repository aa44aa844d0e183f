"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd SQLite file (<name>_results.db).

usage: python tools/rocpd_summary.py RESULTS.db [name-filter-substring ...]
Prints calls / avg / min / max duration (us), grid, workgroup, VGPRs and LDS per (kernel, grid)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main() -> None:
    db = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2:]
    rows = db.execute(
        "select name, grid_x*grid_y*grid_z, workgroup_x, vgpr_count, lds_size, count(*), avg(end-start), min(end-start), max(end-start), "
        "sum(end-start) from kernels group by name, grid_x*grid_y*grid_z order by sum(end-start) desc").fetchall()
    print(f"{'kernel':58s} {'grid':>8s} {'wg':>5s} {'vgpr':>5s} {'lds':>6s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
    for name, grid, wg, vgpr, lds, n, avg, mn, mx, _tot in rows:
        if filt and not any(f in name for f in filt):
            continue
        print(f"{short(name)[:58]:58s} {grid:8d} {wg:5d} {vgpr:5d} {lds:6d} {n:6d} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}")


if __name__ == "__main__":
    main()
