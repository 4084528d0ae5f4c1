#!/usr/bin/env python
"""MFMA utilisation and LDS bank-conflict share per hand-written kernel from rocprofv3 --pmc passes
(MfmaUtil, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE; one pass each, same command as the traffic passes).

usage: python tools/util_summary.py <dir with pmc_MfmaUtil/ pmc_SQ_LDS_BANK_CONFLICT/ pmc_SQ_LDS_IDX_ACTIVE/> > profiles/rNN_pmc_util.json
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMDs) * 100: share of the launch during which the matrix
pipes were busy (any MFMA dtype), averaged over the launches of the kernel.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_summary import collect, NAMES  # noqa: E402

NAMES.append(("hipblaslt_gemm", "Cijk_"))


def main():
    root = sys.argv[1]
    util = collect(os.path.join(root, "pmc_MfmaUtil"), "MfmaUtil")
    conf = collect(os.path.join(root, "pmc_SQ_LDS_BANK_CONFLICT"), "SQ_LDS_BANK_CONFLICT")
    act = collect(os.path.join(root, "pmc_SQ_LDS_IDX_ACTIVE"), "SQ_LDS_IDX_ACTIVE")
    out = {}
    for k in sorted(set(util) | set(conf)):
        u, c, a = util.get(k, []), conf.get(k, []), act.get(k, [])
        out[k] = {"launches_profiled": len(u), "mfma_util_pct": round(sum(u) / max(len(u), 1), 1),
                  "lds_bank_conflict_pct_of_lds_cycles": round(100.0 * sum(c) / sum(a), 1) if a and sum(a) else None}
    json.dump({"source": "rocprofv3 --pmc MfmaUtil / SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE over `bench.py --steps 2 --warmup 1`",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
