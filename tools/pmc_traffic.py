#!/usr/bin/env python3
"""Turn rocprofv3 counter-collection CSVs into profiles/*_pmc_traffic.json (read by bench.py for roofline.traffic).

Collection (separate passes, --kernel-trace only; MI355X_MICROARCH.md "HBM" / "rocprofv3" sections):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- \
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --serial
    rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- \
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --serial
    python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write profiles/rNN_pmc_traffic.json [command] [batch points clouds]
bench.py attaches the file's numbers to a run only when batch / points / clouds match (default 32 16384 uniform).

Correction (gfx950): FETCH_SIZE tallies each 128-byte read request at 64 B, so read bytes = 2 * FETCH_SIZE KiB;
WRITE_SIZE is exact.  bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of each kernel
instantiation.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short_name(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.strip()


def load(directory):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                acc[short_name(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --serial"
    fetch, write = load(fetch_dir), load(write_dir)
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("pdm::"):
            continue
        f = fetch.get(k, {}).get("FETCH_SIZE", [])
        w = write.get(k, {}).get("WRITE_SIZE", [])
        a = write.get(k, {}).get("TCC_EA0_ATOMIC_sum", [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        kernels[k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_mean": round(fm, 1), "WRITE_SIZE_KiB_mean": round(wm, 1),
                      "hbm_bytes_per_launch_corrected": int((2 * fm + wm) * 1024),
                      "TCC_EA0_ATOMIC_mean": round(sum(a) / len(a), 1) if a else 0.0}
    shape = {"batch": int(sys.argv[5]) if len(sys.argv) > 5 else 32, "points": int(sys.argv[6]) if len(sys.argv) > 6 else 16384,
             "clouds": sys.argv[7] if len(sys.argv) > 7 else "uniform"}
    doc = {"shape": shape, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum (separate passes) -- " + cmd,
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
           "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote %s: %d kernels" % (out, len(kernels)))


if __name__ == "__main__":
    main()
