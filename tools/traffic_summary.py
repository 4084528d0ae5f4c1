#!/usr/bin/env python
"""HBM traffic per launch of the hand-written kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: python tools/traffic_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> > profiles/rNN_traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled;
WRITE_SIZE is exact for 16-byte streaming stores.  (The FFT kernels' 8-byte mode accesses are outside the
calibrated pattern: their corrected numbers are indicative.)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NAMES = [("pce_gemm", "pce_kernel"), ("adam", "adam_kernel"), ("irfft", "irfft_split_kernel"), ("irfft", "irfft_kernel"), ("rfft", "rfft_split_kernel"), ("rfft", "rfft_kernel"),
         ("legendre_fwd", "legendre_fwd_x3_kernel"), ("legendre_inv", "legendre_inv_x3_kernel"),
         ("dhconv_fwd", "dhconv_fwd_x3_kernel"), ("dhconv_dgrad", "dhconv_dgrad_x3_kernel"),
         ("dhconv_wgrad", "dhconv_wgrad_x3_kernel"),
         ("legendre_fwd", "legendre_fwd_kernel"), ("legendre_inv", "legendre_inv_kernel"),
         ("dhconv_fwd", "dhconv_fwd_kernel"), ("dhconv_dgrad", "dhconv_dgrad_kernel"),
         ("dhconv_wgrad", "dhconv_wgrad_kernel"), ("conv1x1_wgrad", "conv1x1_wgrad_kernel"), ("conv1x1_wgrad", "conv1x1_wgrad_big_kernel"), ("bias_gelu_fwd", "bias_gelu_fwd_kernel"),
         ("bias_gelu_bwd", "bias_gelu_bwd_kernel"), ("instnorm", "instnorm_"), ("instnorm", "rowsum2_kernel")]


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):   # rocprofv3's default rocpd output
        import sqlite3
        cur = sqlite3.connect(f).cursor()
        for kname, val in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            for short, pat in NAMES:
                if pat in kname:
                    acc[short].append(float(val))
                    break
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for short, pat in NAMES:
                if pat in r["Kernel_Name"]:
                    acc[short].append(float(r["Counter_Value"]))
                    break
    return acc


def main():
    root = sys.argv[1]
    fetch = collect(os.path.join(root, "pmc_FETCH_SIZE"), "FETCH_SIZE")
    write = collect(os.path.join(root, "pmc_WRITE_SIZE"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wr = 1024.0 * sum(w) / max(len(w), 1)
        out[k] = {"launches_profiled": len(f), "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 2 --warmup 1`",
               "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
