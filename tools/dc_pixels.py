#!/usr/bin/env python3
"""tools/dc_pixels.py -- stego pixels of the product path against the fp64 oracle, with and without TFFT_DC_BIAS=128."""
import os, sys, hashlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _checkers import Checker, Params
from steganosaurus_amd import binding as B
from steganosaurus_amd.synth import cover_rgb, n_stream_bits
orc = Checker("orc")
pk = hashlib.sha256(b"test123").digest()
for (w, h, secret) in ((512, 512, 1024), (1024, 1024, 4096), (600, 400, 64)):
    n = n_stream_bits(secret)
    img = cover_rgb(w, h, 3)
    bits = np.random.default_rng(1).integers(0, 2, n).astype(np.uint8)
    want, _, bins = orc.embed_rgb8(img, pk, bits, Params(), want_bins=True)
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    wl = B.Walk(orc.subkeys(pk)[0], ph, pw).next(n)
    for mode in ("0", "128"):
        os.environ["TFFT_DC_BIAS"] = mode
        ctx = B.Context(w, h)
        del os.environ["TFFT_DC_BIAS"]
        ctx.forward_rgb8(img)
        ctx.embed_bins(wl, bits)
        got = ctx.inverse_rgb8(w, h)
        ctx.close()
        d = np.abs(got.astype(int) - want.astype(int))
        print("%dx%d bias %3s: pixels differing from the fp64 reference %d of %d (%.4f %%), max |diff| %d" % (w, h, mode, int((d > 0).sum()), d.size, 100.0 * (d > 0).mean(), int(d.max())))
