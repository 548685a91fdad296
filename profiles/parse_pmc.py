"""Aggregates a rocprofv3 `--pmc` counter_collection.csv by kernel: mean counter value per dispatch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -o NAME -- python bench.py ...
    python profiles/parse_pmc.py DIR/NAME_counter_collection.csv > profiles/rNN_pmc_fetch.json

FETCH_SIZE is in KiB of 64-byte TCC->EA read requests; on gfx950 a wide coalesced stream is
counted at exactly half its bytes (MI355X_MICROARCH.md, HBM section), so
hbm_read_bytes ~= FETCH_SIZE * 1024 * 2.
"""
import csv
import json
import sys
from collections import defaultdict


CROSS_ATTN_BYTES_PER_CHUNK_LAYER = 2 * 1500 * 1280 * 2     # large-v3: K and V^T of one chunk, one layer, fp16
CROSS_ATTN_THREADS_PER_CHUNK = 20 * 512                    # grid (heads, chunks) x 512 threads


def main(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    grid = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for row in rd:
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            cname = row.get("Counter_Name") or row.get("Counter Name") or ""
            val = row.get("Counter_Value") or row.get("Counter Value") or "0"
            short = name.split("(")[0]
            a = acc[short][cname]
            a[0] += float(val)
            a[1] += 1
            g = row.get("Grid_Size") or row.get("Grid_Size_X")
            if g and cname == "FETCH_SIZE":
                grid[short][0] += float(g)
                grid[short][1] += 1
    out = {}
    for k, cs in acc.items():
        out[k] = {c: {"mean": v[0] / max(1, v[1]), "dispatches": v[1]} for c, v in cs.items()}
        if "FETCH_SIZE" in out[k]:
            out[k]["hbm_read_bytes_per_launch_corrected"] = out[k]["FETCH_SIZE"]["mean"] * 1024 * 2
            if grid[k][1]:
                out[k]["grid_threads_mean"] = grid[k][0] / grid[k][1]
                if "dec_cross_attn_kernel" in k:
                    # merged decode runs differ in their chunk count: the launch's grid says how many chunks it streamed
                    # (large-v3 geometry), so the algorithmic bytes of the MEAN launch follow from the mean grid
                    chunks = out[k]["grid_threads_mean"] / CROSS_ATTN_THREADS_PER_CHUNK
                    out[k]["chunks_per_launch_mean"] = chunks
                    out[k]["algorithmic_bytes_per_launch"] = chunks * CROSS_ATTN_BYTES_PER_CHUNK_LAYER
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
