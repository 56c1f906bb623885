#!/usr/bin/env python3
"""tools/audit_diag.py W H [center] -- where the fp32 product spectrum is furthest from the fp64 audit transform."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from steganosaurus_amd import binding as B
from steganosaurus_amd.synth import cover_rgb
w, h = int(sys.argv[1]), int(sys.argv[2]); center = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
img = cover_rgb(w, h, 7)
ctx = B.Context(w, h)
pw, ph = ctx.forward_rgb8(img, center=center)
got = ctx.download_spectrum(pw, ph).astype(np.complex128)
want = ctx.audit_forward_rgb8_f64(img, center=center)
for p in range(3):
    err = np.abs(got[p] - want[p]); rms = np.sqrt(np.mean(np.abs(want[p]) ** 2))
    score = err / (1e-4 * np.abs(want[p]) + 1e-5 * rms)
    off = np.ones(score.shape, bool); off[:, 0] = off[:, pw // 2] = False; off[0, :] = off[ph // 2, :] = False
    peak = max(abs(want[p][0, 0]), abs(want[p][ph // 2, 0]), abs(want[p][0, pw // 2]), abs(want[p][ph // 2, pw // 2]))
    relm = np.where(off & (np.abs(want[p]) >= 0.1 * rms), err / np.maximum(np.abs(want[p]), 1e-300), 0.0)
    iy, ix = np.unravel_index(np.argmax(relm), relm.shape)
    print("plane", p, "worst off-axis pure-relative error among |F| >= 0.1 rms: %.3g at y %d x %d (|F|/rms %.3g)" % (relm[iy, ix], iy, ix, abs(want[p][iy, ix]) / rms))
    print("plane", p, "worst OFF-axis score %.3f; worst on-axis err %.3g*rms = %.3g ulp_fp32(peak)" % (score[off].max(), err[~off].max() / rms, err[~off].max() / (5.97e-8 * peak)))
    idx = np.argsort(score.ravel())[-6:][::-1]
    print("plane", p, "rms %.4g" % rms, "normwise %.3g" % (np.linalg.norm(got[p] - want[p]) / np.linalg.norm(want[p])))
    for i in idx:
        y, x = divmod(int(i), pw)
        print("   y %5d x %5d  |want|/rms %.4g  err/rms %.3g  err/|want| %.3g  score %.3f" % (y, x, abs(want[p][y, x]) / rms, err[y, x] / rms, err[y, x] / abs(want[p][y, x]), score[y, x]))
