#!/usr/bin/env python3
"""Generate tests/golden/cover_hash.json from the REFERENCE itself (oracle/_ref/libtfref.so, the reference TU
compiled in place): compute_cover_hash (S:415-444) as do_embed / do_extract call it.  Build container only:

    make -C oracle all && python tests/gen_golden_coverhash.py

Each case stores the inputs' recipe (seeded synthetic cover, size, centring) and the reference's outputs: region,
the 3*region^2 magnitudes, their quantised bytes and the 32-byte hash.  The last cases are QUANTISATION-EDGE covers
found by a seeded search: one of their magnitudes lies within ~1e-6 (relative) of a bucket edge exp(2k)-1 of
floor(log(1+mag)/2), which is where an fp32 spectrum could land on the other side.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _checkers import Checker  # noqa: E402
from steganosaurus_amd.synth import cover_rgb, gradient_cover  # noqa: E402

EDGES = np.exp(2.0 * np.arange(1, 8)) - 1.0


def edge_distance(mags):
    """smallest relative distance of any magnitude to a quantiser edge"""
    m = np.asarray(mags)[:, None]
    return float((np.abs(m - EDGES[None, :]) / EDGES[None, :]).min())


def search_edge_cases(n_keep=2, tries=60000):
    """seeded search with numpy (same magnitudes as the reference up to ~1e-13): small covers, any centring"""
    rng = np.random.default_rng(415444)
    best = []
    for t in range(tries):
        w, h = int(rng.integers(40, 97)), int(rng.integers(40, 97))
        idx = int(rng.integers(0, 1 << 20))
        center = int(rng.integers(0, 2))
        img = cover_rgb(w, h, idx).astype(np.float64)
        if center:
            yy, xx = np.mgrid[0:h, 0:w]
            img = img * np.where((xx + yy) & 1, -1.0, 1.0)[:, :, None]
        pw, ph = 1 << (w - 1).bit_length(), 1 << (h - 1).bit_length()
        region = min(8, min(ph, pw) // 8)
        pad = np.zeros((ph, pw, 3))
        pad[:h, :w] = img
        # only the region x region corner is needed: two small matrix products
        ey = np.exp(2j * np.pi * np.outer(np.arange(region), np.arange(ph)) / ph)
        ex = np.exp(2j * np.pi * np.outer(np.arange(pw), np.arange(region)) / pw)
        mags = np.abs(np.einsum("yn,nmp,mx->pyx", ey, pad, ex)).ravel()
        d = edge_distance(mags)
        best.append((d, w, h, idx, center))
        best.sort()
        best = best[:n_keep]
    return best


def main():
    R = Checker("ref")
    cases = []

    def add(kind, w, h, index, center, note=""):
        img = cover_rgb(w, h, index) if kind == "lcg" else gradient_cover(w, h, index)
        region, hh, mags, q = R.cover_hash(img, center)
        cases.append({"cover": kind, "w": w, "h": h, "index": index, "center": center, "region": region,
                      "hash": hh.hex(), "q": q.tolist(), "mags": [float(v) for v in mags],
                      "edge_distance": edge_distance(mags) if len(mags) else None, "note": note})
        print(kind, w, h, index, center, region, hh.hex()[:16], cases[-1]["edge_distance"], flush=True)

    add("lcg", 256, 256, 0, 0)
    add("lcg", 256, 256, 0, 1)
    add("grad", 256, 256, 1, 0)
    add("lcg", 512, 512, 0, 0)
    add("lcg", 512, 512, 2, 1)
    add("lcg", 600, 400, 0, 0, "pads to 1024x512")
    add("lcg", 300, 500, 1, 1, "pads to 512x512")
    add("lcg", 48, 40, 0, 1, "pads to 64x64")
    add("lcg", 16, 8, 0, 0, "region 1")
    add("lcg", 5, 3, 0, 0, "region 0: hash of the empty string")
    add("lcg", 1920, 1080, 0, 0, "BASELINE configs[1] geometry")
    for d, w, h, idx, center in search_edge_cases():
        add("lcg", w, h, idx, center, "quantisation edge (search distance %.2e)" % d)
    with open(os.path.join(HERE, "golden", "cover_hash.json"), "w") as f:
        json.dump({"source": "oracle/_ref/libtfref.so ref_cover_hash = compute_cover_hash S:415-444", "cases": cases}, f, indent=0)


if __name__ == "__main__":
    main()
