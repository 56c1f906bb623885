"""Parity checks of the HIP path against the oracle / golden vectors, written once
and run twice: on a real MI355X through libturtlefft_hip.so (tests/test_gpu_parity.py,
-m gpu) and on the CPU-emulated build of the same kernel sources
(tests/test_emulated.py), where they check index math only.

Bars (north_star): extracted bits bit-exact; FFT coefficients within 1e-4 relative.
"""
import hashlib
import os

import numpy as np
import pytest

from _checkers import Params, bins_digest
from steganosaurus_amd import binding as B
from steganosaurus_amd.synth import cover_rgb, gradient_cover, secret_ascii, n_stream_bits

PASS = "test123"
PK = hashlib.sha256(PASS.encode()).digest()


def spec_errors(got, want):
    """How "FFT coefficients within 1e-4 relative" (north_star) is measured for an fp32
    transform checked against the fp64 reference.  Returns
      nrm : ||got-want|| / ||want||                       (normwise relative error)
      mx  : max over ALL coefficients of |got-want| / (rtol*|want| + atol) with
            OFF the axes  rtol = 1e-5, atol = 1e-6*rms(|want|)   -- ten times inside the north_star
            tolerance; the library takes the mean term out of the pixels before the transform and adds its
            exact transform back (DC removal, on by default), so no partial sum carries it;
            ON the excluded axes x in {0,PW/2}, y in {0,PH/2} (S:698-700: never embedded or read)
            rtol = 1e-4, atol = max(4e-5*rms, 2 ulp_fp32 of the DC-like peak): the true values there are as
            large as the peak itself, and 0.3-0.8 ulp of the peak is the fp32 representation limit of the
            spectrum on those rows/columns.
            <= 1 means every coefficient is inside its bar.
      rel : max |got-want| / |want| over OFF-AXIS coefficients with |want| >= 0.1 * rms
    A per-coefficient relative figure is only meaningful for coefficients that are not
    far below the spectrum's rms: an fp32 FFT has an absolute rounding floor of a few
    1e-7 * rms on every output, whatever that output's own size."""
    want = np.asarray(want)
    got = np.asarray(got).astype(np.complex128)
    err = np.abs(got - want)
    nrm = np.linalg.norm(got - want) / max(1e-300, np.linalg.norm(want))
    rms = max(1e-300, np.sqrt(np.mean(np.abs(want) ** 2)))
    big = np.abs(want) >= 0.1 * rms
    if want.ndim == 2:
        big[:, 0] = big[:, want.shape[1] // 2] = False
        big[0, :] = big[want.shape[0] // 2, :] = False
    rel = (err[big] / np.abs(want)[big]).max() if big.any() else 0.0
    rtol = np.full(want.shape, 1e-5)
    atol = np.full(want.shape, 1e-6 * rms)
    if want.ndim == 2:
        ph, pw = want.shape
        peak = max(abs(want[0, 0]), abs(want[ph // 2, 0]), abs(want[0, pw // 2]), abs(want[ph // 2, pw // 2]))
        axis = max(4e-5 * rms, 2 * 5.97e-8 * peak)
        atol[:, 0] = atol[:, pw // 2] = axis
        atol[0, :] = atol[ph // 2, :] = axis
        rtol[:, 0] = rtol[:, pw // 2] = 1e-4
        rtol[0, :] = rtol[ph // 2, :] = 1e-4
    return nrm, (err / (rtol * np.abs(want) + atol)).max(), rel


def assert_spectrum_close(got, want, tag=""):
    nrm, mx, rel = spec_errors(got, want)
    assert nrm < 1e-6, (tag, "normwise", nrm)            # 100x inside the 1e-4 tolerance
    assert mx <= 1.0, (tag, "off-axis rtol 1e-5 + 1e-6*rms / on-axis rtol 1e-4 + peak ulps", mx)
    assert rel < 1e-4, (tag, "per-coefficient", rel)     # the north_star tolerance, never widened


def check_fft_kat(lib):
    """delta at (y=1,x=1) on 4x8: F[0][1] = (+.7071,+.7071), F[1][0] = (0,+1) (SURVEY finding 3)."""
    img = np.zeros((4, 8, 3), np.uint8)
    img[1, 1, :] = 1
    ctx = B.Context(8, 4, lib=lib)
    pw, ph = ctx.forward_rgb8(img)
    assert (pw, ph) == (8, 4)
    F = ctx.download_spectrum(pw, ph)
    for p in range(3):
        assert abs(F[p, 0, 1] - (0.70710678 + 0.70710678j)) < 1e-6
        assert abs(F[p, 1, 0] - 1j) < 1e-6
    ctx.close()


def check_forward_against_oracle(lib, orc, sizes, centers=(0, 1)):
    for (w, h) in sizes:
        img = cover_rgb(w, h, 0)
        ctx = B.Context(w, h, lib=lib)
        for center in centers:
            want, med = orc.forward_rgb8(img, center)
            pw, ph = ctx.forward_rgb8(img, center)
            assert (ph, pw) == want.shape[1:]
            got = ctx.download_spectrum(pw, ph)
            for p in range(3):
                assert_spectrum_close(got[p], want[p], (w, h, center, p))
            # medians and capacity are EXACT against the fp64 reference (tfft_exact.hip: the bins within the fp32 error of the decision
            # value are re-evaluated in fp64 from the pixels): values to 1e-12 relative, the count as an integer
            # (a one-pixel-wide image, whose internal row length 2 is not the reference's 1, keeps the fp32 statistics and their bars)
            gm = ctx.medians()
            exact = w >= 2
            assert (not exact) or all(n > 0 for n in ctx.exact_info()), ("fp64 refinement did not run", w, h, ctx.exact_info())
            assert np.allclose(gm, med, rtol=1e-12 if exact else 2e-6, atol=0), (w, h, gm, med)
            cap_want, _ = orc.capacity_rgb8(img, Params(center=center))
            cap = ctx.capacity(0.01 * gm)
            assert (cap == cap_want) if exact else (abs(cap - cap_want) <= 2), (w, h, cap, cap_want)
        ctx.close()


def check_forward_golden(lib, golden_dir, wh, center):
    g = np.load(os.path.join(golden_dir, f"fft_{wh[0]}x{wh[1]}_c{center}.npz"))
    img = cover_rgb(wh[0], wh[1], int(g["cover_index"]))
    ctx = B.Context(wh[0], wh[1], lib=lib)
    pw, ph = ctx.forward_rgb8(img, center)
    got = ctx.download_spectrum(pw, ph)
    for p in range(3):
        assert_spectrum_close(got[p], g["spec"][p], (wh, center, p))
    assert np.allclose(ctx.medians(), g["med"], rtol=1e-12, atol=0)
    ctx.close()


def check_median_paths(lib, orc, sizes):
    """The sampled fast path and the forced full fallback give the same exact order statistic; on
    ordinary covers the fast path is the one that runs."""
    for (w, h) in sizes:
        img = cover_rgb(w, h, 4)
        _, want = orc.forward_rgb8(img, want_spec=False)
        res = {}
        for mode in ("0", "1"):
            os.environ["TFFT_MEDIAN_FALLBACK"] = mode
            try:
                ctx = B.Context(w, h, lib=lib)
            finally:
                os.environ.pop("TFFT_MEDIAN_FALLBACK", None)
            ctx.forward_rgb8(img)
            res[mode] = (ctx.medians().copy(), ctx.median_path().copy())
            ctx.close()
        assert np.array_equal(res["0"][0], res["1"][0]), (w, h, res)
        assert np.allclose(res["0"][0], want, rtol=1e-12, atol=0)
        assert res["1"][1].sum() == 0
        if w * h >= 64 * 64:
            assert res["0"][1].sum() == 3, ("fast path not taken", w, h, res["0"][1])
    # degenerate spectrum (constant image: every AC magnitude ~0): whatever path runs, the answer is exact
    flat = np.full((32, 32, 3), 77, np.uint8)
    ctx = B.Context(32, 32, lib=lib)
    ctx.forward_rgb8(flat)
    m = ctx.medians()
    _, want = orc.forward_rgb8(flat, want_spec=False)
    assert np.all(np.abs(m - want) <= 1e-3), (m, want)
    ctx.close()


def check_context_reuse(lib, orc, max_wh, sizes):
    """One context, many geometries in sequence (plans switch between direct / two-step / fused columns, big
    then small images reuse the same pools): every result equals the one from a fresh context."""
    ctx = B.Context(max_wh[0], max_wh[1], slots=2, lib=lib)
    rng = np.random.default_rng(9)
    for (w, h) in sizes:
        img = cover_rgb(w, h, 5)
        ph, pw = orc.next_pow2(h), orc.next_pow2(w)
        n = 120
        try:
            bins = B.Walk(orc.subkeys(PK)[0], ph, pw, lib=lib).next(n)
        except B.TfftError:
            bins = None                      # too small for 120 positions: transform-only check
        bits = rng.integers(0, 2, n).astype(np.uint8)
        res = []
        for c in (ctx, B.Context(w, h, lib=lib)):
            slot = 1 if c is ctx else 0
            c.forward_rgb8(img, slot=slot)
            spec = c.download_spectrum(pw, ph, slot=slot)
            med = c.medians(slot=slot)
            if bins is not None:
                c.embed_bins(bins, bits, slot=slot)
            out = c.inverse_rgb8(w, h, slot=slot)
            res.append((spec, med, out))
            if c is not ctx:
                c.close()
        assert np.array_equal(res[0][0], res[1][0]), (w, h)
        assert np.array_equal(res[0][1], res[1][1]), (w, h)
        assert np.array_equal(res[0][2], res[1][2]), (w, h)
    ctx.close()


def check_bit_index(lib, orc, w, h, n, jitter=0.0):
    """tfft_bins_sort + tfft_set_bit_index: visiting the bins in address order leaves every caller-visible
    result identical (spectrum after embed, stego bytes, extracted bits in stream order), with and without
    jitter; an index that is not a permutation or is used with another n is refused."""
    img = cover_rgb(w, h, 11)
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    keys = orc.subkeys(PK)
    bins = B.Walk(keys[0], ph, pw, lib=lib).next(n)
    jit = B.walk_jitter(b"".join(keys[1:4]), bins, jitter, lib=lib) if jitter else None
    bits = np.random.default_rng(21).integers(0, 2, n).astype(np.uint8)
    sbins, idx = B.bins_sort(bins, lib=lib)
    key = (sbins["plane"].astype(np.int64) << 32) | (sbins["y"].astype(np.int64) << 16) | sbins["x"]
    assert (np.diff(key) > 0).all() and np.array_equal(sbins, bins[idx]) and np.array_equal(np.sort(idx), np.arange(n))
    res = []
    for ordered in (False, True):
        ctx = B.Context(w, h, lib=lib)
        if ordered:
            ctx.set_bit_index(idx)
        bl = sbins if ordered else bins
        ctx.forward_rgb8(img)
        ctx.embed_bins(bl, bits, jitter=jit)
        spec = ctx.download_spectrum(pw, ph)
        stego = ctx.inverse_rgb8(w, h)
        ctx.forward_rgb8(stego)
        got = ctx.read_bins(bl, jitter=jit)
        res.append((spec, stego, got))
        if ordered:
            with pytest.raises(B.TfftError):            # index of n entries, call with n - 1 bins
                ctx.read_bins(bl[:-1])
            bad = idx.copy(); bad[0] = bad[1]
            with pytest.raises(B.TfftError):
                ctx.set_bit_index(bad)
            bad = idx.copy(); bad[0] = n
            with pytest.raises(B.TfftError):
                ctx.set_bit_index(bad)
            ctx.set_bit_index(None)                     # cleared: plain walk order works again
            assert np.array_equal(ctx.read_bins(bins, jitter=jit), got)
        ctx.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    return res[0][2], bits


def check_audit64_against_oracle(lib, orc, sizes):
    """(f-4) the fp64 audit transform is the reference's fft2d bit for bit: forward of u8 images (padded,
    with and without centring), and forward + inverse of seeded complex planes."""
    ctx = B.Context(64, 64, lib=lib)
    rng = np.random.default_rng(64)
    for (w, h) in sizes:
        img = cover_rgb(w, h, 2)
        for center in (0, 1):
            want = orc.forward_rgb8(img, center=bool(center))[0]
            got = ctx.audit_forward_rgb8_f64(img, center=bool(center))
            assert got.shape == want.shape and np.array_equal(got.view(np.float64), want.view(np.float64)), (w, h, center)
        ph, pw = orc.next_pow2(h), orc.next_pow2(w)
        z = (rng.standard_normal((2, ph, pw)) + 1j * rng.standard_normal((2, ph, pw))) * 100
        for inverse in (False, True):
            want = np.stack([orc.fft2d(z[i], inverse=inverse) for i in range(2)])
            got = ctx.audit_fft2d_f64(z, inverse=inverse)
            assert np.array_equal(got.view(np.float64), want.view(np.float64)), (w, h, inverse)
    ctx.close()


def check_product_against_audit64(lib, w, h, center=False, seed=7):
    """fp32 product spectrum against the fp64 audit transform of the same image, at any size (no CPU
    reference needed): the SURVEY 8c tolerance (1e-4 relative, see spec_errors)."""
    img = cover_rgb(w, h, seed)
    ctx = B.Context(w, h, lib=lib)
    pw, ph = ctx.forward_rgb8(img, center=center)
    got = ctx.download_spectrum(pw, ph)
    want = ctx.audit_forward_rgb8_f64(img, center=center)
    ctx.close()
    for p in range(3):
        assert_spectrum_close(got[p], want[p], tag="%dx%d plane %d" % (w, h, p))
    return got, want


def check_dc_removal(lib, sizes, ref_forward):
    """DC removal (the default; TFFT_DC_BIAS=0 switches it off): the forward transform of (pixel - 128) plus the
    analytic transform of the constant.  Against the fp64 reference (ref_forward(img, center) -> complex128
    spectrum) every coefficient is inside the standard bars (off-axis 1e-5*|F| + 1e-6*rms), forward -> inverse
    returns the image.  With the switch OFF (A/B measurements only) the transform still round-trips exactly and stays
    normwise < 2e-6, but heavily padded images (2131x1179 -> 4096x2048) then only reach ~1.5e-4 beside the excluded axes
    (2e-4*|F| + 2e-5*rms is asserted): that is the reason the removal is on by default."""
    for i, (w, h) in enumerate(sizes):
        img = cover_rgb(w, h, 40 + i)
        center = bool(i & 1)
        want = ref_forward(img, center)
        for mode in ("on", "off"):
            if mode == "off":
                os.environ["TFFT_DC_BIAS"] = "0"
            try:
                ctx = B.Context(w, h, lib=lib)
            finally:
                os.environ.pop("TFFT_DC_BIAS", None)
            pw, ph = ctx.forward_rgb8(img, center=center)
            got = ctx.download_spectrum(pw, ph).astype(np.complex128)
            assert np.array_equal(ctx.inverse_rgb8(w, h), img), (w, h, mode)
            ctx.close()
            for p in range(3):
                if mode == "on":
                    assert_spectrum_close(got[p], want[p], (w, h, p, "dc removal on"))
                else:
                    err = np.abs(got[p] - want[p]); rms = max(1e-300, np.sqrt(np.mean(np.abs(want[p]) ** 2)))
                    off = np.ones(err.shape, bool); off[:, 0] = off[0, :] = False
                    off[:, pw // 2] = False; off[ph // 2, :] = False
                    if off.any():
                        score = (err / (2e-4 * np.abs(want[p]) + 2e-5 * rms))[off].max()
                        assert score <= 1.0, (w, h, p, "off-axis, dc removal off", score)
                    assert np.linalg.norm(got[p] - want[p]) / np.linalg.norm(want[p]) < 2e-6, (w, h, p, "dc removal off")


def check_identity_roundtrip(lib, sizes):
    """forward -> inverse with no embedding returns the cover exactly (integers survive fp32)."""
    for (w, h) in sizes:
        img = cover_rgb(w, h, 1)
        ctx = B.Context(w, h, lib=lib)
        for center in (0, 1):
            ctx.forward_rgb8(img, center)
            out = ctx.inverse_rgb8(w, h)
            assert np.array_equal(out, img), (w, h, center, int((out != img).sum()))
        ctx.close()


def check_walk_against_oracle(lib, orc, cases):
    kw = orc.subkeys(PK)[0]
    for (ph, pw, n, rmin, rmax, dens) in cases:
        rc, want, sk, ctr, start = orc.walk(kw, ph, pw, n, rmin, rmax, dens)
        assert rc == 0
        wk = B.Walk(kw, ph, pw, rmin, rmax, dens, lib=lib)
        assert list(wk.start()) == start.tolist()
        # resumable: two chunks == one call
        a = wk.next(n // 3)
        b = wk.next(n - n // 3)
        got = B.bins_to_triples(np.concatenate([a, b]))
        assert np.array_equal(got, want), (ph, pw)
        assert wk.skipped == sk and wk.ks_blocks() == ctr
        wk.close()


def check_embed_extract(lib, orc, w, h, n_bits, params_kw, seed=3, gradient=False, spectrum_bars=True):
    """GPU embed vs oracle embed on the same cover/bits; GPU read of the oracle's
    stego vs oracle's raw bits (bit-exact); and the full GPU->GPU round trip."""
    P = Params(**params_kw)
    img = gradient_cover(w, h, seed) if gradient else cover_rgb(w, h, seed)
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, n_bits).astype(np.uint8)
    want_stego, want_spec, want_bins = orc.embed_rgb8(img, PK, bits, P, want_spec=True, want_bins=True)
    want_raw = orc.extract_bits(want_stego, PK, n_bits, P)
    sub = orc.subkeys(PK)
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)

    wk = B.Walk(sub[0], ph, pw, P.rmin, P.rmax, P.density, lib=lib)
    bins = wk.next(n_bits)
    assert np.array_equal(B.bins_to_triples(bins), want_bins)
    jit = B.walk_jitter(sub[1] + sub[2] + sub[3], bins, P.jitter, lib=lib) if P.jitter != 0.0 else None

    ctx = B.Context(w, h, lib=lib)
    ctx.forward_rgb8(img, P.center)
    med = ctx.medians()
    ctx.embed_bins(bins, bits, P.alpha, jit, P.adaptive_alpha, med)
    got_spec = ctx.download_spectrum(pw, ph)
    for p in range(3):
        if spectrum_bars:
            assert_spectrum_close(got_spec[p], want_spec[p], ("after-embed", w, h, p))
        else:       # DC removal switched off: the north_star tolerance with the old floors
            e = np.abs(got_spec[p] - want_spec[p]); rms = np.sqrt(np.mean(np.abs(want_spec[p]) ** 2))
            assert np.linalg.norm(e) / np.linalg.norm(want_spec[p]) < 2e-6
    stego = ctx.inverse_rgb8(w, h)
    diff = stego.astype(np.int16) - want_stego
    assert np.abs(diff).max() <= 1, ("stego differs by more than 1 LSB", np.abs(diff).max())
    frac = float((diff != 0).mean())
    assert frac < 0.01, ("fraction of +-1 LSB pixels vs the fp64 reference", frac)

    # GPU read of the ORACLE's stego: raw bits identical to the oracle's raw bits
    ctx.forward_rgb8(want_stego, P.center)
    med2 = ctx.medians()
    got_raw = ctx.read_bins(bins, P.alpha, jit, P.adaptive_alpha, med2)
    bad = np.nonzero(got_raw != want_raw)[0]
    if len(bad):
        # a disagreement is only tolerable where the reference's own decision is a coin flip
        spec2, _ = orc.forward_rgb8(want_stego, P.center)
        t = B.bins_to_triples(bins[bad])
        v = spec2[t[:, 0], t[:, 1], t[:, 2]]
        assert P.jitter == 0 and not P.adaptive_alpha and np.all(np.abs(v.imag) < 1e-5 * np.abs(v)), (len(bad), v[:4])
    # GPU -> GPU round trip: same bit-error pattern class as the reference (pow2: few errors)
    ctx.forward_rgb8(stego, P.center)
    rt = ctx.read_bins(bins, P.alpha, jit, P.adaptive_alpha, ctx.medians())
    ctx.close()
    return {"ber_gpu": float((rt != bits).mean()), "ber_ref": float((want_raw != bits).mean()), "lsb_frac": frac}


def check_config1_golden(lib, orc, golden_dir, name):
    """BASELINE.json configs[0]: 512x512, 1 KB secret, defaults; bits from the reference's own framing."""
    g = np.load(os.path.join(golden_dir, f"embed_512_{name}.npz"))
    img = cover_rgb(512, 512, 0) if name == "lcg" else gradient_cover(512, 512, 1)
    n = int(g["n_bits"])
    bits = np.unpackbits(g["bits"])[:n]
    ref_stego = (img.astype(np.int16) + g["stego_diff"]).astype(np.uint8)
    ref_raw = np.unpackbits(g["raw"])[:n]
    sub = orc.subkeys(PK)
    wk = B.Walk(sub[0], 512, 512, lib=lib)
    bins = wk.next(n)
    assert bins_digest(B.bins_to_triples(bins)) == str(g["bins_sha256"])
    ctx = B.Context(512, 512, lib=lib)
    # extract side on the reference-made stego: raw bits identical
    ctx.forward_rgb8(ref_stego)
    got = ctx.read_bins(bins)
    assert np.array_equal(got, ref_raw), int((got != ref_raw).sum())
    # embed side
    ctx.forward_rgb8(img)
    ctx.embed_bins(bins, bits)
    stego = ctx.inverse_rgb8(512, 512)
    d = stego.astype(np.int16) - ref_stego
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 0.01, (np.abs(d).max(), (d != 0).mean())
    # the oracle (= reference) extracts the GPU-made stego: payload survives ECC exactly like the reference's
    back = orc.extract_bits(stego, PK, n)
    ctx.close()
    return back, bits, ref_raw


def check_error_paths(lib):
    ctx = B.Context(64, 64, lib=lib)
    try:
        ctx.medians()
        raise AssertionError("medians before forward must fail")
    except B.TfftError as e:
        assert e.status == -6
    try:
        ctx.forward_rgb8(np.zeros((65, 64, 3), np.uint8))
        raise AssertionError("oversize image must fail")
    except B.TfftError as e:
        assert e.status == -3
    ctx.forward_rgb8(np.zeros((64, 64, 3), np.uint8))
    bad = B.make_bins(np.array([[0, 0, 5]]))         # y == 0: excluded axis
    try:
        ctx.embed_bins(bad, np.array([1], np.uint8))
        raise AssertionError("axis bin must be rejected")
    except B.TfftError as e:
        assert e.status == -8
    # a payload length whose bit count wraps in 64 bits (38*24 + 56*len == 912 + 56 mod 2^64) must not pass for a short stream
    dummy = np.zeros(64, np.uint8)
    for plen in ((1 << 64) // 56 + 2, (1 << 63)):
        try:
            ctx.embed_stream_batch_dev(1, dummy.ctypes.data, 64, 64, dummy.ctypes.data, 2000, dummy.ctypes.data, dummy.ctypes.data, plen, dummy.ctypes.data)
            raise AssertionError("payload length beyond the walk must be rejected")
        except B.TfftError as e:
            assert e.status == -1, e.status      # TFFT_E_INVALID
    ctx.close()
    try:
        B.Context(70000, 8, lib=lib)
        raise AssertionError("too large")
    except B.TfftError as e:
        assert e.status == -3


def load_cover_hash_cases(golden_dir, max_pixels=None):
    import json
    with open(os.path.join(golden_dir, "cover_hash.json")) as f:
        cases = json.load(f)["cases"]
    return [c for c in cases if max_pixels is None or c["w"] * c["h"] <= max_pixels]


def cover_of(case):
    return cover_rgb(case["w"], case["h"], case["index"]) if case["cover"] == "lcg" else gradient_cover(case["w"], case["h"], case["index"])


def host_cover_hash(host, mags):
    """the CLI's quantiser + SHA-256 (tf_frame.cpp cover_hash_from_mags, S:433-443) through libtfhost.so"""
    import ctypes as C
    m = np.ascontiguousarray(mags, np.float64).ravel()
    out = C.create_string_buffer(32)
    host.tfh_cover_hash_from_mags(m.ctypes.data_as(C.c_void_p), C.c_size_t(m.size), out)
    return out.raw


def check_cover_hash(lib, host, golden_dir, max_pixels=None):
    """(f-2) compute_cover_hash S:415-444: tfft_lowfreq_mag (fp64 inner products with the pixels) -> the CLI's
    quantiser -> SHA-256 equals the REFERENCE's 32-byte hash on every fixture, including the covers whose
    magnitudes sit 2e-7 / 4e-7 (relative) from a quantiser edge; magnitudes within 1e-11 of the reference's; and
    they agree with the resident fp32 spectrum (same bins) to the transform's own accuracy."""
    for c in load_cover_hash_cases(golden_dir, max_pixels):
        img = cover_of(c)
        ctx = B.Context(c["w"], c["h"], lib=lib)
        pw, ph = ctx.forward_rgb8(img, c["center"])
        region = min(8, min(ph, pw) // 8)
        assert region == c["region"], c
        if region == 0:
            assert host_cover_hash(host, np.zeros(0)).hex() == c["hash"]
            with pytest.raises(B.TfftError):
                ctx.lowfreq_mag(1 if min(ph, pw) < 1 else 9)        # region out of range is refused
            ctx.close()
            continue
        mags = ctx.lowfreq_mag(region)
        want = np.array(c["mags"]).reshape(3, region, region)
        assert np.all(np.abs(mags - want) <= 1e-11 * np.maximum(want, 1.0)), (c["w"], c["h"], np.abs(mags - want).max())
        assert host_cover_hash(host, mags).hex() == c["hash"], (c["w"], c["h"], c["note"])
        spec = np.abs(ctx.download_spectrum(pw, ph)[:, :region, :region].astype(np.complex128))
        rms = np.sqrt(np.mean(want ** 2))
        assert np.all(np.abs(spec - want) <= 1e-4 * want + 1e-5 * rms + 2 * 5.97e-8 * want.max()), (c["w"], c["h"])
        ctx.close()


class HostBufs:
    """device buffers of the emulated runtime are host arrays"""
    @staticmethod
    def put(a):
        a = np.ascontiguousarray(a).copy()
        return a, a.ctypes.data

    @staticmethod
    def get(h):
        return h


class TorchBufs:
    """device buffers on cuda:0 through torch (plumbing only)"""
    @staticmethod
    def put(a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        torch.cuda.synchronize()
        return t, t.data_ptr()

    @staticmethod
    def get(h):
        return h.cpu().numpy()


def make_header(clen, salt_seed=0, ver=2):
    """Header::to_bytes S:886-904: FTTG, ver, flags=0, salt16, nonce12, clen_be32"""
    rng = np.random.default_rng(1000 + salt_seed)
    return np.frombuffer(b"FTTG" + bytes([ver, 0]) + rng.bytes(16) + rng.bytes(12) + int(clen).to_bytes(4, "big"), np.uint8).copy()


def rep_stream(header, payload):
    """bits_from_bytes + rep3 / rep7 (S:455-467, S:494-500, S:986-995) in numpy"""
    return np.concatenate([np.repeat(np.unpackbits(header), 3), np.repeat(np.unpackbits(payload), 7)])


def check_stream_batch(lib, orc, bufs, w, h, secrets=(40, 40, 40, 100, 100), slots=3, sort=True):
    """tfft_embed_stream_batch_dev / tfft_extract_stream_batch_dev: packed header + payload bytes in, packed bytes out,
    the length learnt from the image (S:1223-1264).  Power-of-two cover, so every byte comes back."""
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    assert (ph, pw) == (h, w), "use a power-of-two cover: others do not round-trip in the reference either"
    max_plen = max(secrets) + 16
    n_bins = 912 + 56 * max_plen + 500                       # the extractor's walk: longer than any stream it will meet
    bins = B.Walk(orc.subkeys(PK)[0], ph, pw, lib=lib).next(n_bins)
    ctx = B.Context(w, h, slots=slots, lib=lib)
    ubins = bins
    if sort:
        ubins, idx = B.bins_sort(bins, lib=lib)
        ctx.set_bit_index(idx)
    kb, pb = bufs.put(ubins.view(np.uint8).reshape(-1, 8))
    rng = np.random.default_rng(77)
    groups = {}
    for i, sl in enumerate(secrets):
        groups.setdefault(sl, []).append(i)
    nimg = len(secrets)
    covers = np.stack([cover_rgb(w, h, 50 + i) for i in range(nimg)])
    headers = np.stack([make_header(sl, i) for i, sl in enumerate(secrets)])
    payloads = [rng.integers(0, 256, sl + 16).astype(np.uint8) for sl in secrets]
    stego = np.zeros_like(covers)
    for sl, members in groups.items():                        # one embed call per payload length
        plen = sl + 16
        ci, cp = bufs.put(covers[members]); hi, hp = bufs.put(headers[members])
        pi, pp = bufs.put(np.stack([payloads[m] for m in members])); oi, op = bufs.put(np.zeros_like(covers[members]))
        ui, up = bufs.put(np.zeros(len(members), np.int64))
        ctx.embed_stream_batch_dev(len(members), cp, w, h, pb, n_bins, hp, pp, plen, op, usable_ptr=up)
        ctx.sync()
        out = bufs.get(oi)
        # == the bit-level call fed with the numpy expansion of the same bytes (bits beyond the stream: never written)
        bits = np.zeros((len(members), n_bins), np.uint8)
        for k, m in enumerate(members):
            st = rep_stream(headers[m], payloads[m]); bits[k, :len(st)] = st
        # positions >= the stream length must stay untouched: embed bit-level with a list cut to the stream length
        n_bits = 912 + 56 * plen
        c2 = B.Context(w, h, slots=slots, lib=lib)
        cut = bins[:n_bits]
        if sort:
            cut, idx2 = B.bins_sort(cut, lib=lib); c2.set_bit_index(idx2)
        k2, p2 = bufs.put(cut.view(np.uint8).reshape(-1, 8)); b2, bp2 = bufs.put(bits[:, :n_bits]); o2, op2 = bufs.put(np.zeros_like(covers[members]))
        c2.embed_batch_dev(len(members), cp, w, h, p2, bp2, n_bits, op2)
        c2.sync()
        assert np.array_equal(bufs.get(o2), out), ("stream embed != bit-level embed", sl)
        c2.close()
        assert (bufs.get(ui) >= n_bits).all()
        for k, m in enumerate(members):
            stego[m] = out[k]
    # one extraction call over images of different payload lengths + a plain cover + a future version + a too-long clen
    bad_ver = stego[0].copy(); too_long = stego[0].copy()
    c3 = B.Context(w, h, lib=lib)
    for arr, hdr in ((bad_ver, make_header(secrets[0], 0, ver=3)), (too_long, make_header(10 ** 6, 0))):
        st = rep_stream(hdr, payloads[0])
        c3.forward_rgb8(covers[0]); c3.embed_bins(bins[:len(st)], st); arr[...] = c3.inverse_rgb8(w, h)
    c3.close()
    batch = np.concatenate([stego, covers[:1], bad_ver[None], too_long[None]])
    nb = len(batch)
    bi, bp = bufs.put(batch); ho, hop = bufs.put(np.zeros((nb, 38), np.uint8)); po, pop = bufs.put(np.zeros((nb, max_plen), np.uint8))
    so, sop = bufs.put(np.zeros(nb, np.int32)); ro, rop = bufs.put(np.zeros((nb, n_bins), np.uint8))
    ctx.extract_stream_batch_dev(nb, bp, w, h, pb, n_bins, hop, pop, max_plen, sop, raw_bits_out_ptr=rop)
    ctx.sync()
    status, hdr_out, pay_out, raw = bufs.get(so), bufs.get(ho), bufs.get(po), bufs.get(ro)
    assert status[:nimg].tolist() == list(secrets), status
    assert status[nimg:].tolist() == [-1, -2, -3], status
    for i in range(nimg):
        assert np.array_equal(hdr_out[i], headers[i]), i
        assert np.array_equal(pay_out[i, :secrets[i] + 16], payloads[i]), i
    assert hdr_out[nimg + 1][4] == 3
    # raw bits == the bit-level extraction of the same batch; without raw_bits_out the decoded bytes are the same
    r2, rp2 = bufs.put(np.zeros((nb, n_bins), np.uint8))
    ctx.extract_batch_dev(nb, bp, w, h, pb, n_bins, rp2)
    ctx.sync()
    assert np.array_equal(bufs.get(r2), raw)
    h3, hp3 = bufs.put(np.zeros((nb, 38), np.uint8)); p3, pp3 = bufs.put(np.zeros((nb, max_plen), np.uint8)); s3, sp3 = bufs.put(np.zeros(nb, np.int32))
    ctx.extract_stream_batch_dev(nb, bp, w, h, pb, n_bins, hp3, pp3, max_plen, sp3)
    ctx.sync()
    assert np.array_equal(bufs.get(s3), status) and np.array_equal(bufs.get(h3), hdr_out) and np.array_equal(bufs.get(p3), pay_out)
    # a registered bin list (tfft_bins_register_dev): what extraction derives from it is kept between calls -- same results, twice
    ctx.bins_register_dev(pb, n_bins)
    for _ in range(2):
        r3, rp3 = bufs.put(np.zeros((nb, n_bins), np.uint8))
        ctx.extract_batch_dev(nb, bp, w, h, pb, n_bins, rp3)
        ctx.sync()
        assert np.array_equal(bufs.get(r3), raw)
    ctx.bins_register_dev(None, 0)
    # a walk shorter than the stream is refused at embed time; a walk shorter than what the header announces is status -3
    with pytest.raises(B.TfftError):
        ctx.embed_stream_batch_dev(1, bp, w, h, pb, n_bins, hop, pop, max_plen + 100, bp)
    # the HOST-buffer variants (three-stream pipeline, packed bytes over PCIe) return the same bytes
    members = groups[secrets[0]]
    h_out = np.zeros_like(covers[members]); h_us = np.zeros(len(members), np.uint64)
    ctx.embed_stream_batch_host(np.ascontiguousarray(covers[members]), ubins, np.ascontiguousarray(headers[members]),
                                np.stack([payloads[m] for m in members]), h_out, usable=h_us)
    assert np.array_equal(h_out, stego[members]) and (h_us >= 912 + 56 * (secrets[0] + 16)).all()
    hh = np.zeros((nb, 38), np.uint8); hp = np.zeros((nb, max_plen), np.uint8); hs = np.zeros(nb, np.int32); hr = np.zeros((nb, n_bins), np.uint8)
    ctx.extract_stream_batch_host(batch, ubins, hh, hp, hs, raw_bits_out=hr)
    assert np.array_equal(hs, status) and np.array_equal(hh, hdr_out) and np.array_equal(hr, raw)
    for i in range(nimg):
        assert np.array_equal(hp[i, :secrets[i] + 16], payloads[i]), i
    ctx.close()


def _ctx_with_env(env, *a, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return B.Context(*a, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def check_delta_embedding(lib, orc, bufs, w, h, n_bits, nimg=2, rmax=0.45, center=False, sort=True, lsb_frac=0.01, with_oracle=True):
    """Batched embedding as the default pipeline runs it -- stego = cover + IFFT(F' - F), the first inverse column step building its
    tiles from the bucketed bins (S:712-732 per bin, S:1099-1102 by linearity) -- against (a) the fp64 reference's stego image,
    (b) the write-F'-then-invert pipeline (TFFT_EMBED_DELTA=0), and (c) the reference's reading of OUR stego image."""
    P = Params(rmax=rmax, center=int(center))
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    covers = np.stack([cover_rgb(w, h, 70 + i) for i in range(nimg)])
    bits = np.random.default_rng(5).integers(0, 2, (nimg, n_bits)).astype(np.uint8)
    bins = B.Walk(orc.subkeys(PK)[0], ph, pw, rmin=P.rmin, rmax=rmax, lib=lib).next(n_bits)
    assert len(bins) == n_bits
    ubins, idx = (B.bins_sort(bins, lib=lib) if sort else (bins, None))
    kb, pb = bufs.put(ubins.view(np.uint8).reshape(-1, 8))
    cb, pc = bufs.put(covers)
    bb, pbits = bufs.put(bits)
    out = {}
    for mode in ("1", "0"):
        ctx = _ctx_with_env({"TFFT_EMBED_DELTA": mode}, w, h, slots=max(1, nimg - 1), lib=lib)      # two chunks when nimg > 1
        if idx is not None:
            ctx.set_bit_index(idx)
        ob, po = bufs.put(np.zeros_like(covers))
        usable = np.zeros(nimg, np.uint64)
        ub, pu = bufs.put(usable)
        ctx.embed_batch_dev(nimg, pc, w, h, pb, pbits, n_bits, po, center=center, rmax=rmax, usable_ptr=pu)
        ctx.sync()
        out[mode] = (bufs.get(ob).copy(), bufs.get(ub).copy())
        if mode == "1":      # the raw bits of our stego images, read by the batched extraction
            rb, pr = bufs.put(np.zeros((nimg, n_bits), np.uint8))
            ctx.extract_batch_dev(nimg, po, w, h, pb, n_bits, pr, center=center)
            ctx.sync()
            raw = bufs.get(rb).copy()
        ctx.close()
    (sd, ud), (s0, u0) = out["1"], out["0"]
    # embedding in place (output buffer = cover buffer): the last kernel reads the cover's bytes it is about to overwrite
    ctx = B.Context(w, h, slots=max(1, nimg - 1), lib=lib)
    if idx is not None:
        ctx.set_bit_index(idx)
    ib, pi = bufs.put(covers)
    ctx.embed_batch_dev(nimg, pi, w, h, pb, pbits, n_bits, pi, center=center, rmax=rmax)
    ctx.sync()
    assert np.array_equal(bufs.get(ib), sd), "in-place embedding"
    ctx.close()
    # TFFT_STATS_SKEW (test hook): every bracket moved off the median -- the fast path of the statistics fails and their fallbacks
    # (on the |F|^2 planes the delta pipeline stores) must return the same capacities
    # ... TFFT_STATS_M2=0 / TFFT_STATS_ASYNC=0: the statistics on the complex spectrum, in line (the A/B forms of the default)
    # ... TFFT_STATS_TILE=0: the |F|^2 planes + the statistics kernels over them instead of the classification inside the last forward
    # column step (the default since round 3); with the skew the gated fallback of either form runs
    # ... TFFT_STATS_TILE=2: the in-kernel form also for launches of few images (by default from 2^24 bins per launch on)
    envs = [{"TFFT_STATS_SKEW": "5"}, {"TFFT_STATS_TILE": "0", "TFFT_STATS_M2": "0", "TFFT_STATS_ASYNC": "0"}]
    pi = B.Context(w, h, slots=max(1, nimg - 1), lib=lib)
    plan = pi.plan_info(w, h, max(1, nimg - 1)); pi.close()
    if plan["two_step"] and 4 <= plan["log_n2"] <= 9 and (max(2, pw) // 2) % 16 == 0:      # the in-kernel form can serve this geometry
        envs += [{"TFFT_STATS_TILE": "2"}, {"TFFT_STATS_TILE": "2", "TFFT_STATS_SKEW": "5"}]
        if max(1, nimg - 1) * ph * max(2, pw) >= (1 << 24):      # ... and is the default at this launch size: the planes' form as the variant
            envs += [{"TFFT_STATS_TILE": "0"}, {"TFFT_STATS_TILE": "0", "TFFT_STATS_SKEW": "5"}]
    for env in envs:
        ctx = _ctx_with_env(env, w, h, slots=max(1, nimg - 1), lib=lib)
        if idx is not None:
            ctx.set_bit_index(idx)
        ob, po = bufs.put(np.zeros_like(covers)); ub, pu = bufs.put(np.zeros(nimg, np.uint64))
        ctx.embed_batch_dev(nimg, pc, w, h, pb, pbits, n_bits, po, center=center, rmax=rmax, usable_ptr=pu)
        ctx.sync()
        assert np.array_equal(bufs.get(ub), ud) and np.array_equal(bufs.get(ob), sd), ("statistics variants of the delta pipeline", env)
        ctx.close()
    assert np.array_equal(ud, u0)
    assert np.array_equal(bufs.get(cb), covers), "the cover buffer is read, never written"
    stats = []
    for i in range(nimg):
        dm = np.abs(sd[i].astype(np.int16) - s0[i])
        assert dm.max() <= 1 and float((dm != 0).mean()) < lsb_frac, (i, dm.max(), float((dm != 0).mean()))
        assert np.any(sd[i] != covers[i])
        if not with_oracle:      # sizes the fp64 reference needs minutes for: the two pipelines against each other, and the round trip
            assert float((raw[i] != bits[i]).mean()) < 0.02 or (ph, pw) != (h, w)
            stats.append((None, float((dm != 0).mean())))
            continue
        want = orc.embed_rgb8(covers[i], PK, bits[i], P)[0]
        dd = sd[i].astype(np.int16) - want
        d0 = s0[i].astype(np.int16) - want
        assert np.abs(dd).max() <= 1, ("delta stego differs from the fp64 reference by more than 1 LSB", i, np.abs(dd).max())
        fd, f0 = float((dd != 0).mean()), float((d0 != 0).mean())
        assert fd < lsb_frac, (i, fd)
        assert np.abs(sd[i].astype(np.int16) - s0[i]).max() <= 1
        stats.append((fd, f0))
        want_raw = orc.extract_bits(sd[i], PK, n_bits, P)
        bad = np.nonzero(raw[i] != want_raw)[0]
        if len(bad):      # tolerable only where the reference's own decision is a coin flip
            spec2, _ = orc.forward_rgb8(sd[i], P.center)
            t = B.bins_to_triples(bins[bad])
            v = spec2[t[:, 0], t[:, 1], t[:, 2]]
            assert np.all(np.abs(v.imag) < 1e-5 * np.abs(v)), (i, len(bad), v[:4])
    return stats


def check_batch_capacity(lib, bufs, w, h, nimg=3, cases=((0.05, 0.45, 0.01), (0.0, 1.5, 0.3), (0.1, 0.6, 1.0), (0.2, 0.3, 2.5), (0.05, 0.45, 0.0)), flat=False,
                         envs=({}, {"TFFT_STATS_TILE": "2"}, {"TFFT_STATS_TILE": "0"}, {"TFFT_MEDIAN_FALLBACK": "1"}, {"TFFT_STATS_FUSED": "0"})):
    """Capacity counted inside the medians' full pass (batch path, S:998-1008) == tfft_capacity with thr = magmin * median,
    image by image, exactly: default annulus, an annulus that reaches into the mirror half (rmax > 0.5), thresholds at and
    above the median (thousands of bins inside the threshold's bracket: the parked list overflows and the plain kernel
    recounts), magmin = 0, and the forced median fallback."""
    imgs = np.stack([cover_rgb(w, h, 70 + i) for i in range(nimg)])
    if flat:      # a flat image: every bin but a few has the same magnitude -- all of them land in the median's bracket (the in-kernel
        imgs[0][:] = 77      # statistics stage a tile's candidates in 192 LDS slots per wave: here every tile overflows them)
    ph, pw = 1 << (h - 1).bit_length(), 1 << (w - 1).bit_length()
    bins = B.Walk(bytes(range(32)), ph, pw, 0.0, 1.5, 0.9, lib=lib).next(16)
    bits = np.ones((nimg, 16), np.uint8)
    ii, ip = bufs.put(imgs); ki, kp = bufs.put(bins.view(np.uint8).reshape(-1, 8)); bi, bp = bufs.put(bits)
    oi, op = bufs.put(np.zeros_like(imgs))
    one = B.Context(w, h, lib=lib)
    want = {}
    for (rmin, rmax, magmin) in cases:
        for i in range(nimg):
            one.forward_rgb8(imgs[i])
            want[(rmin, rmax, magmin, i)] = one.capacity(magmin * one.medians(), rmin, rmax)
    one.close()
    for env in envs:
        os.environ.update(env)
        try:
            ctx = B.Context(w, h, slots=2, lib=lib)
        finally:
            for k in env:
                del os.environ[k]
        for (rmin, rmax, magmin) in cases:
            ui, up = bufs.put(np.full(nimg, -1, np.int64))
            ctx.embed_batch_dev(nimg, ip, w, h, kp, bp, 16, op, rmin=rmin, rmax=rmax, magmin=magmin, usable_ptr=up)
            ctx.sync()
            got = bufs.get(ui)
            for i in range(nimg):
                assert int(got[i]) == want[(rmin, rmax, magmin, i)], (env, rmin, rmax, magmin, i, int(got[i]), want[(rmin, rmax, magmin, i)])
        ctx.close()
    # a statistics sequence that broke off (test hook: garbage in the select state + an error) must not poison the next call
    os.environ["TFFT_STATS_FAIL_ONCE"] = "1"
    try:
        ctx = B.Context(w, h, slots=2, lib=lib)
    finally:
        del os.environ["TFFT_STATS_FAIL_ONCE"]
    (rmin, rmax, magmin) = cases[0]
    ui, up = bufs.put(np.full(nimg, -1, np.int64))
    try:
        ctx.embed_batch_dev(nimg, ip, w, h, kp, bp, 16, op, rmin=rmin, rmax=rmax, magmin=magmin, usable_ptr=up)
        raise AssertionError("the injected failure did not surface")
    except B.TfftError:
        pass
    ctx.embed_batch_dev(nimg, ip, w, h, kp, bp, 16, op, rmin=rmin, rmax=rmax, magmin=magmin, usable_ptr=up)
    ctx.sync()
    got = bufs.get(ui)
    for i in range(nimg):
        assert int(got[i]) == want[(rmin, rmax, magmin, i)], ("after a broken statistics sequence", i, int(got[i]), want[(rmin, rmax, magmin, i)])
    ctx.close()

