#!/usr/bin/env python3
"""tools/cmp_tile.py W H [n_bits] -- stego bytes and capacities of the batched embed with the statistics inside the last forward column
step (default) against TFFT_STATS_TILE=0 (|F|^2 planes) and against no statistics at all; prints where they differ"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from steganosaurus_amd import binding as B
from steganosaurus_amd.synth import cover_rgb
w, h = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
nimg = 4
ph = 1 << (h - 1).bit_length(); pw = 1 << (w - 1).bit_length()
bins = B.Walk(bytes(range(32)), ph, pw).next(n)
sb, idx = B.bins_sort(bins)
dev = torch.device("cuda:0")
covers = np.stack([cover_rgb(w, h, i) for i in range(nimg)])
d_img = torch.from_numpy(covers).to(dev)
bits = np.random.default_rng(1).integers(0, 2, (nimg, n), dtype=np.uint8)
d_bits = torch.from_numpy(bits).to(dev)
d_bins = torch.from_numpy(sb.view(np.uint8).reshape(-1, 8).copy()).to(dev)
res = {}
VARIANTS = [("tile", {}, True), ("planes", {"TFFT_STATS_TILE": "0"}, True), ("nostats", {}, False), ("tile2", {}, True)]
for ph in os.environ.get("CMP_PH", "").split(","):
    if ph:
        VARIANTS.append(("ph" + ph, {"TFFT_DBG_PH": ph}, True))
for name, env, us in VARIANTS:
    for k, v in env.items():
        os.environ[k] = v
    ctx = B.Context(w, h, slots=nimg)
    for k in env:
        os.environ.pop(k)
    ctx.set_bit_index(idx)
    d_out = torch.zeros_like(d_img); d_us = torch.zeros(nimg, dtype=torch.int64, device=dev)
    for rep in range(2):
        ctx.embed_batch_dev(nimg, d_img.data_ptr(), w, h, d_bins.data_ptr(), d_bits.data_ptr(), n, d_out.data_ptr(), usable_ptr=d_us.data_ptr() if us else None)
        ctx.sync()
    res[name] = (d_out.cpu().numpy(), d_us.cpu().numpy())
    ctx.close()
for a in [v[0] for v in VARIANTS[1:]]:
    d = res["tile"][0] != res[a][0]
    print("tile vs", a, ": differing bytes", int(d.sum()), "capacities", res["tile"][1], res[a][1])
    if d.any():
        i, y, x, ch = np.argwhere(d)[0]
        ys = np.unique(np.argwhere(d)[:, 1]); xs = np.unique(np.argwhere(d)[:, 2])
        dd = res["tile"][0].astype(int) - res[a][0].astype(int)
        print("  per image", [int(d[k].sum()) for k in range(nimg)], "values", np.unique(dd[d]))
        print("  first at image %d (y %d, x %d, ch %d); rows %s.. cols %s.. images %s" % (i, y, x, ch, ys[:8], xs[:8], np.unique(np.argwhere(d)[:, 0])))
