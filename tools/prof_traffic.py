#!/usr/bin/env python3
"""tools/prof_traffic.py <summary.json> <workload> [traffic.json] -- HBM bytes per launch per bench stage
from the PMC means of tools/prof_summary.py: (2*FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE doubled on gfx950,
MI355X_MICROARCH.md, HBM/rocprofv3 section).  Stages made of several kernels sum their kernels
(launches per stage given below).  Updates profiles/traffic.json in place."""
import json, os, sys

summary, workload = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
S = json.load(open(summary))


def kib(name):
    v = S.get(name)
    if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        return None
    return 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]


def first(*names):
    for n in names:
        b = kib(n)
        if b is not None:
            return b
    return None


def prefix(*prefixes):
    """first kernel (template arguments after the prefix ignored: DC / TW switches) that has counters"""
    for p in prefixes:
        for name in S:
            if name.startswith(p):
                b = kib(name)
                if b is not None:
                    return b
    return None


stages = {
    "rows_fwd": first("k_rowcol_fwd<3>", "k_rows_fwd<11, 1>", "k_rows_fwd<12, 1>", "k_rows_fwd<10, 3>"),
    "rows_inv": first("k_colrow_inv<3>", "k_rows_inv<11, 1>", "k_rows_inv<12, 1>", "k_rows_inv<10, 3>"),
    "cols_fwd_b": prefix("k_fft_cols<8, 1, 0", "k_fft_cols<6, 1, 0", "k_fft_cols<7, 1, 0"),
    "cols_fwd_read": prefix("k_fft_cols<8, 1, 2", "k_fft_cols<6, 1, 2", "k_fft_cols<7, 1, 2", "k_fft_cols<8, 1, 1", "k_fft_cols<6, 1, 1", "k_fft_cols<7, 1, 1"),
    "cols_inv_a": prefix("k_fft_cols<8, -1, 0", "k_fft_cols<6, -1, 0", "k_fft_cols<7, -1, 0"),
    "embed": kib("k_embed"),
    "read": kib("k_read"),
    "capacity": first("k_capacity<false>", "k_capacity<true>"),
}
med = [kib(k) for k in ("k_collect_bracket",)] + [2 * (kib("k_hist_spec") or 0), kib("k_hist_cand<true>") or 0]
if med[0] is not None:
    stages["medians"] = sum(med)
T = json.load(open(out)) if os.path.exists(out) else {}
T[workload] = {k: int(v * 1024) for k, v in stages.items() if v is not None}
T["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of "
              "`bench.py --batched-only` (tools/prof.sh, tools/prof_traffic.py); the per-kernel means are in profiles/r1/*_pmc_summary.json")
json.dump(T, open(out, "w"), indent=1)
print(json.dumps(T[workload], indent=1))
