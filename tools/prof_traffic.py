#!/usr/bin/env python3
"""tools/prof_traffic.py <summary.json> <workload> <source tag> [traffic.json] -- HBM bytes per launch per bench stage
from the PMC means of tools/prof_summary.py: (2*FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE doubled on gfx950,
MI355X_MICROARCH.md, HBM/rocprofv3 section; calibrated on these kernels' own compulsory byte counts).  Stages made of
several kernels sum their kernels.  Updates profiles/traffic.json in place (one entry per workload + where it came from)."""
import json, os, re, sys

summary, workload, source = sys.argv[1], sys.argv[2], sys.argv[3]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
S = json.load(open(summary))


def kib(name):
    v = S.get(name)
    if not v or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        return None
    return 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]


def match(pattern):
    """the busiest kernel (by share of the run) whose name matches"""
    best = None
    for name, v in S.items():
        if re.match(pattern, name) and kib(name) is not None and (best is None or v["pct"] > S[best]["pct"]):
            best = name
    return (kib(best), best) if best else (None, None)


def match_sum(pattern):
    """a stage that is several launches of sibling kernels (the live-rows fused forward: one per live-row count)"""
    names = [n for n in S if re.match(pattern, n) and kib(n) is not None]
    return (sum(kib(n) for n in names), " + ".join(sorted(names))) if names else (None, None)


stages, kernels = {}, {}
b, name = match_sum(r"k_rowcol_fwd_live<")
if b is not None:
    stages["rows_fwd"] = int(b * 1024)
    kernels["rows_fwd"] = name
for stage, pat in {
    "rows_fwd": r"k_rowcol_fwd<|k_rows_fwd<",
    "rows_inv": r"k_colrow_inv<|k_rows_inv<",
    # k_fft_cols<LOGL, SIGN, MODE, DC, TW, FULL>
    "cols_fwd_a": r"k_fft_cols<\d+, 1, 0, false, true, (true|false)>",                 # first forward column step (output twiddles)
    "cols_fwd_b": r"k_fft_cols<\d+, 1, [45], (true|false), false, (true|false)>|k_fft_cols<\d+, 1, 0, (true|false), false, (true|false)>",      # final forward column step (mode 4 / 5: it also writes the listed bins' values, delta embedding; 5 classifies its values for the statistics; the busiest match wins over the sample pass, mode 0)
    "cols_fwd_read": r"k_fft_cols<\d+, 1, [12], ",                       # the same step as extraction runs it
    "cols_inv_a": r"k_fft_cols<\d+, -1, [03], (true|false), true, (true|false)>",      # first inverse column step (mode 3: tiles built from the bin lists, delta embedding)
    "cols_inv_b": r"k_fft_cols<\d+, -1, 0, false, false, (true|false)>",               # last inverse column step (three-pass plans)
    "embed": r"k_gather_bits$|k_embed$", "read": r"k_read$", "capacity": r"k_capacity<",
}.items():
    if stage in stages:
        continue
    b, name = match(pat)
    if b is not None:
        stages[stage] = int(b * 1024)
        kernels[stage] = name
med = [kib(k) for k in S if k.startswith("k_collect_bracket")] + [kib(k) or 0 for k in ("k_hist_spec", "k_hist_cand<true>", "k_hist_cand2", "k_col0_stats")]
if any(re.match(r"k_fft_cols<\d+, 1, 5, ", k) for k in S):      # statistics inside the last forward step: their sample pass is the plain last step over every 8th tile
    med += [kib(k) or 0 for k in S if re.match(r"k_fft_cols<\d+, 1, 0, (true|false), false, true>", k)]
med = [m for m in med if m is not None]
if med:
    stages["medians"] = int(sum(med) * 1024)
T = json.load(open(out)) if os.path.exists(out) else {}
T[workload] = stages
T.setdefault("_kernels", {})[workload] = kernels
T.setdefault("_source", {})[workload] = source
try:      # the counters are per launch: remember how many images a launch of that run covered
    bt = json.load(open(os.path.join(os.path.dirname(summary), "bench_trace.json")))
    T.setdefault("_images_per_launch", {})[workload] = bt["config"]["images_per_launch"]
except Exception:
    pass
T["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of `bench.py --batched-only "
              "--workload <w>` (tools/prof.sh, tools/prof_traffic.py); per-kernel means in profiles/<round>/*_pmc_summary.json.  bench.py "
              "replays these numbers as roofline.traffic and says so")
json.dump(T, open(out, "w"), indent=1)
print(json.dumps(stages, indent=1))
