#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes:
the TCC block has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md).

Units/corrections as that guide prescribes: both counters are in KiB (x1024); on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled before being
compared with byte counts; WRITE_SIZE is exact for 16-byte-per-lane stores.

  python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*counter_collection.csv \
         gpurun_out/pmc_write/*/*counter_collection.csv > profiles/r01_..._pmc_traffic.json
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return agg


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("void fn2::") and not k.startswith("fn2::"):
            continue
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        n = max(nf, nw, 1)
        out[k.replace("void ", "")] = {
            "launches": n,
            "fetch_bytes_per_launch_raw": f * 1024 / max(nf, 1),
            "fetch_bytes_per_launch_x2": 2 * f * 1024 / max(nf, 1),
            "write_bytes_per_launch": w * 1024 / max(nw, 1),
            "hbm_bytes_per_launch": (2 * f * 1024 / max(nf, 1)) + (w * 1024 / max(nw, 1)),
        }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
