"""FETCH_SIZE per kernel of a rocprofv3 --pmc run, in dispatch order groups: for A/B micro-benchmarks that alternate
variants of ONE kernel inside a process (profiles/attn_bench.py, gemm_bench.py) the per-kernel mean of parse_pmc.py would
mix the variants; this prints, per kernel name and per grid size, the mean over consecutive runs of equal dispatches.
FETCH_SIZE is KiB of 64-byte requests and counts a wide stream at half its bytes on gfx950 (x2 applied here)."""
import csv
import sys
from itertools import groupby


def main(path):
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if (row.get("Counter_Name") or "") != "FETCH_SIZE":
                continue
            name = (row.get("Kernel_Name") or "").split("(")[0]
            grid = row.get("Grid_Size") or row.get("Grid_Size_X") or ""
            rows.append((int(row.get("Dispatch_Id") or 0), name, grid, float(row.get("Counter_Value") or 0)))
    rows.sort()
    # one dispatch may be split over XCC rows: sum per dispatch id
    per = {}
    for d, n, g, v in rows:
        k = per.setdefault(d, [n, g, 0.0])
        k[2] += v
    seq = [per[d] for d in sorted(per)]
    for (n, g), grp in groupby(seq, key=lambda r: (r[0], r[1])):
        vals = [r[2] for r in grp]
        mb = sum(vals) / len(vals) * 1024 * 2 / 1e6
        print(f"{n[:70]:70s} grid {g:>8s}  x{len(vals):3d}  FETCH {mb:9.1f} MB per launch (x2-corrected)")


if __name__ == "__main__":
    main(sys.argv[1])
