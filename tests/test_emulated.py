"""CPU-emulated run of the HIP kernel sources (tests/emu): checks index math,
barrier placement and the C-ABI host logic without a GPU.  This is NOT the
product path and proves nothing about the gfx950 build -- tests/test_gpu_parity.py
(-m gpu) is the parity gate.  Sizes are tiny: one fiber per HIP thread."""
import os
import subprocess

import numpy as np
import pytest

import parity_cases as PC
from steganosaurus_amd import binding as B

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", EMU_DIR], check=True, stdout=subprocess.DEVNULL)
    return B.load(os.path.join(EMU_DIR, "libtfft_emu.so"))


def test_fft_sign_kat(emu):
    PC.check_fft_kat(emu)


def test_forward_small(emu, orc):
    PC.check_forward_against_oracle(emu, orc, [(64, 64), (48, 40), (16, 8), (2, 2), (3, 1), (1, 5), (100, 30)])


def test_forward_two_step_columns(emu, orc):
    # PH = 1024 > 256 takes the two-step (N1 x N2) column path
    PC.check_forward_against_oracle(emu, orc, [(8, 1024), (20, 600)], centers=(0,))


def test_forward_wide_rows_wave_sync_path(emu, orc):
    # PW = 2048 -> M = 1024 = 64 lanes x 16 elements: the row passes run wave-synchronously (no s_barrier)
    PC.check_forward_against_oracle(emu, orc, [(1500, 3), (2048, 2)], centers=(0, 1))
    PC.check_identity_roundtrip(emu, [(1500, 3)])
    # PW = 4096 -> M = 2048: one plane per workgroup (two waves), own fast staging path; 2101 is not a multiple of 4
    PC.check_forward_against_oracle(emu, orc, [(2100, 2), (2101, 2)], centers=(0, 1))
    PC.check_identity_roundtrip(emu, [(2100, 3), (2101, 2)])
    # H a multiple of 8: the XCD-aware (row, plane) workgroup order of the one-plane kernels
    PC.check_forward_against_oracle(emu, orc, [(2100, 8)], centers=(1,))
    PC.check_identity_roundtrip(emu, [(2104, 16)])


def test_median_fast_and_fallback_paths(emu, orc):
    PC.check_median_paths(emu, orc, [(64, 64), (48, 40), (8, 4), (128, 32)])


def test_fused_rows_plus_column_step(emu, orc):
    # PW = 2048 and PH >= 128: k_rowcol_fwd (rows + length-8 column step in one kernel), then step B
    PC.check_forward_against_oracle(emu, orc, [(1500, 130), (2047, 129)], centers=(0, 1))
    PC.check_identity_roundtrip(emu, [(1500, 130)])


def test_columns_of_512_with_lds_twiddles(emu, orc):
    """PH = 1024 forced into 2 x 512: the L = 512 column kernels keep their inter-pass twiddles in LDS (all other lengths: registers);
    forward, inverse (output twiddles + DC prologue) and the tile-resident read."""
    os.environ["TFFT_COLS_LOG_N1"] = "1"
    try:
        PC.check_forward_against_oracle(emu, orc, [(40, 1000)], centers=(1,))
        PC.check_identity_roundtrip(emu, [(24, 900)])
        PC.check_embed_extract(emu, orc, 64, 1024, 300, dict(rmin=0.0, rmax=1.5, density=0.9))
    finally:
        del os.environ["TFFT_COLS_LOG_N1"]


def test_fused_live_row_counts(emu, orc):
    """The live-rows-only fused kernels at both widths for every shape of a group: H below N2 (one slab, groups without any live row
    write zeros), H = a multiple of N2 (one launch), NL = 8 (nothing padded), and two different counts in one image."""
    os.environ["TFFT_FUSE_WIDE"] = "2"
    try:
        PC.check_forward_against_oracle(emu, orc, [(1200, 128), (1030, 96)], centers=(1,))
        PC.check_forward_against_oracle(emu, orc, [(2100, 136)], centers=(0,))
        PC.check_identity_roundtrip(emu, [(2047, 128), (1500, 200), (2100, 256)])
    finally:
        del os.environ["TFFT_FUSE_WIDE"]
    os.environ["TFFT_FUSE_LIVE"] = "0"          # the all-rows kernels stay correct (A/B path)
    try:
        PC.check_forward_against_oracle(emu, orc, [(1500, 130)], centers=(1,))
        PC.check_identity_roundtrip(emu, [(1500, 130)])
    finally:
        del os.environ["TFFT_FUSE_LIVE"]


def test_fused_rows_plus_column_step_4096_wide(emu, orc):
    # PW = 4096 and 128 <= PH <= 4096: the same fused kernels with two waves per row (workgroup barriers inside the row transform,
    # which the waves of padded rows sit out): live rows only in the first rows of a workgroup, odd width, centring
    os.environ["TFFT_FUSE_WIDE"] = "2"          # also for single images (the default keeps them on the three-pass plan)
    try:
        PC.check_forward_against_oracle(emu, orc, [(2500, 130)], centers=(0,))
        PC.check_forward_against_oracle(emu, orc, [(4095, 200)], centers=(1,))
        PC.check_identity_roundtrip(emu, [(3000, 140)])
    finally:
        del os.environ["TFFT_FUSE_WIDE"]


def test_identity_roundtrip(emu):
    PC.check_identity_roundtrip(emu, [(64, 64), (48, 40), (33, 17), (2, 2), (1, 1), (5, 1), (1, 7), (12, 1024)])


def test_walk(emu, orc):
    PC.check_walk_against_oracle(emu, orc, [(64, 64, 300, 0.05, 0.45, 0.7), (256, 256, 2480, 0.05, 0.45, 0.7),
                                            (128, 256, 900, 0.1, 0.6, 0.5), (256, 128, 900, 0.0, 1.0, 0.9)])


@pytest.mark.parametrize("kw", [dict(), dict(jitter=0.05), dict(adaptive_alpha=1), dict(center=1),
                                dict(alpha=0.3, density=0.5, rmin=0.1, rmax=0.6)])
def test_embed_extract_64(emu, orc, kw):
    r = PC.check_embed_extract(emu, orc, 64, 64, 300, kw)
    assert r["ber_gpu"] <= r["ber_ref"] + 0.02


def test_embed_extract_nonpow2(emu, orc):
    r = PC.check_embed_extract(emu, orc, 48, 40, 300, dict())
    assert 0.2 < r["ber_gpu"] < 0.5      # the reference cannot round-trip non-pow2 images either (finding 1)


def test_embed_mirror_half(emu, orc):
    # rmax = 1.0 reaches bins with x > PW/2: exercised through the conjugate mirror of the half spectrum
    PC.check_embed_extract(emu, orc, 64, 32, 200, dict(rmin=0.0, rmax=1.5, density=0.9))


def test_error_paths(emu):
    PC.check_error_paths(emu)


def test_batch_matches_single(emu, orc):
    """tfft_*_batch_dev over chunks of slots == the single-image calls (emulated: device pointers are host arrays)."""
    from steganosaurus_amd.synth import cover_rgb, n_stream_bits
    w, h, nimg = 40, 24, 5
    n = 120
    imgs = np.stack([cover_rgb(w, h, i) for i in range(nimg)])
    bits = np.random.default_rng(2).integers(0, 2, (nimg, n)).astype(np.uint8)
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    bins = B.Walk(orc.subkeys(PC.PK)[0], ph, pw, lib=emu).next(n)
    out = np.zeros_like(imgs)
    usable = np.zeros(nimg, np.uint64)
    raw = np.zeros((nimg, n), np.uint8)
    os.environ["TFFT_TILE_READ"] = "3"                   # tile-resident extraction also for chunks of < 8 images
    try:
        ctx = B.Context(w, h, slots=3, lib=emu)
    finally:
        del os.environ["TFFT_TILE_READ"]
    ctx.embed_batch_dev(nimg, imgs.ctypes.data, w, h, bins.ctypes.data, bits.ctypes.data, n, out.ctypes.data,
                        usable_ptr=usable.ctypes.data)
    ctx.extract_batch_dev(nimg, out.ctypes.data, w, h, bins.ctypes.data, n, raw.ctypes.data)
    ctx.sync()
    one = B.Context(w, h, lib=emu)
    for i in range(nimg):
        one.forward_rgb8(imgs[i])
        med = one.medians()
        assert one.capacity(0.01 * med) == int(usable[i])
        one.embed_bins(bins, bits[i])
        st = one.inverse_rgb8(w, h)
        # the single-image calls write F' into the spectrum and invert it, the batch embeds cover + IFFT(F' - F): one image in exact
        # arithmetic, 1 LSB apart on a few pixels in fp32 (check_delta_embedding holds both to the fp64 reference)
        dd = np.abs(st.astype(np.int16) - out[i])
        assert dd.max() <= 1 and (dd != 0).mean() < 0.02, (i, dd.max(), (dd != 0).mean())
        one.forward_rgb8(out[i])
        assert np.array_equal(one.read_bins(bins), raw[i]), i
    one.close()
    # the same batch with the bins in address order (tfft_bins_sort + tfft_set_bit_index) and, for the
    # extraction, the row-limited final column step: identical stego bytes, capacities and stream-order bits
    sbins, idx = B.bins_sort(bins, lib=emu)
    ctx.set_bit_index(idx)
    out2 = np.zeros_like(imgs); usable2 = np.zeros(nimg, np.uint64); raw2 = np.zeros((nimg, n), np.uint8)
    ctx.embed_batch_dev(nimg, imgs.ctypes.data, w, h, sbins.ctypes.data, bits.ctypes.data, n, out2.ctypes.data,
                        usable_ptr=usable2.ctypes.data)
    ctx.extract_batch_dev(nimg, out2.ctypes.data, w, h, sbins.ctypes.data, n, raw2.ctypes.data)
    ctx.sync()
    assert np.array_equal(out2, out) and np.array_equal(usable2, usable) and np.array_equal(raw2, raw)
    # extraction reads the bits out of the LDS-resident column tiles (no spectrum); TFFT_TILE_READ=0 keeps the
    # row-limited spectrum + k_read: same bits, also on the generic path (alpha outside (0, pi))
    os.environ["TFFT_TILE_READ"] = "0"
    try:
        old = B.Context(w, h, slots=3, lib=emu)
    finally:
        del os.environ["TFFT_TILE_READ"]
    old.set_bit_index(idx)
    for alpha in (0.5, 3.5):
        ra = np.zeros((nimg, n), np.uint8); rb = np.full((nimg, n), 7, np.uint8)
        ctx.extract_batch_dev(nimg, out2.ctypes.data, w, h, sbins.ctypes.data, n, ra.ctypes.data, alpha=alpha)
        old.extract_batch_dev(nimg, out2.ctypes.data, w, h, sbins.ctypes.data, n, rb.ctypes.data, alpha=alpha)
        ctx.sync(); old.sync()
        assert np.array_equal(ra, rb), alpha
        if alpha == 0.5:
            assert np.array_equal(ra, raw)
    old.close(); ctx.close()
    os.environ["TFFT_TILE_READ"] = "2"                   # the bucket build without LDS histograms (huge grids)
    try:
        g2 = B.Context(w, h, slots=3, lib=emu)
    finally:
        del os.environ["TFFT_TILE_READ"]
    rc = np.zeros((nimg, n), np.uint8)
    g2.extract_batch_dev(nimg, out.ctypes.data, w, h, bins.ctypes.data, n, rc.ctypes.data)
    g2.sync()
    assert np.array_equal(rc, raw)
    g2.close()


def test_unaligned_device_pointers(emu):
    """Row kernels read/write the u8 rows as aligned 32-bit words: images at odd addresses, odd widths and
    the first/last word of the whole batch (which straddles the buffer edge) must still be exact."""
    from steganosaurus_amd.synth import cover_rgb
    for (w, h, off) in [(7, 5, 1), (33, 6, 3), (40, 9, 2), (5, 3, 0)]:
        img = cover_rgb(w, h, 2)
        raw = np.zeros(img.size + 8, np.uint8)
        raw[off:off + img.size] = img.ravel()
        outraw = np.zeros(img.size + 8, np.uint8)
        ctx = B.Context(w, h, lib=emu)
        ctx.forward_rgb8_dev(raw.ctypes.data + off, w, h)
        ctx.inverse_rgb8_dev(outraw.ctypes.data + off)
        ctx.sync()
        assert np.array_equal(outraw[off:off + img.size].reshape(h, w, 3), img), (w, h, off)
        assert outraw[:off].sum() == 0 and outraw[off + img.size:].sum() == 0      # nothing written outside
        ctx.close()


def test_host_batch_pipeline_matches_device_batch(emu, orc):
    """tfft_embed_batch / tfft_extract_batch (double-buffered halves, copy streams) == the resident-batch calls."""
    from steganosaurus_amd.synth import cover_rgb
    w, h, nimg, n = 40, 24, 7, 100
    imgs = np.stack([cover_rgb(w, h, i) for i in range(nimg)])
    bits = np.random.default_rng(3).integers(0, 2, (nimg, n)).astype(np.uint8)
    bins = B.Walk(orc.subkeys(PC.PK)[0], orc.next_pow2(h), orc.next_pow2(w), lib=emu).next(n)
    ref_out = np.zeros_like(imgs); ref_raw = np.zeros((nimg, n), np.uint8); ref_us = np.zeros(nimg, np.uint64)
    ctx = B.Context(w, h, slots=4, lib=emu)
    ctx.embed_batch_dev(nimg, imgs.ctypes.data, w, h, bins.ctypes.data, bits.ctypes.data, n, ref_out.ctypes.data,
                        usable_ptr=ref_us.ctypes.data)
    ctx.extract_batch_dev(nimg, ref_out.ctypes.data, w, h, bins.ctypes.data, n, ref_raw.ctypes.data)
    ctx.sync()
    for slots in (4, 1, 3):
        c2 = B.Context(w, h, slots=slots, lib=emu)
        out = np.zeros_like(imgs); raw = np.zeros((nimg, n), np.uint8); us = np.zeros(nimg, np.uint64)
        c2.embed_batch_host(imgs, bins, bits, out, usable=us)
        c2.extract_batch_host(out, bins, raw)
        assert np.array_equal(out, ref_out) and np.array_equal(raw, ref_raw) and np.array_equal(us, ref_us), slots
        c2.close()
    ctx.close()


def test_frame_expand_and_majority_against_reference_frames(emu, golden_dir):
    """Device Rep-3/Rep-7 framing == the reference's own frames (kat.json), and majority decode absorbs bit errors."""
    import json
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    ctx = B.Context(8, 8, lib=emu)
    for fr in kat["frames"]:
        want = np.unpackbits(np.frombuffer(bytes.fromhex(fr["bits_packed"]), np.uint8))
        plen = len(fr["secret"]) + 16
        n = 912 + 56 * plen
        want = want[:n]
        hdr = np.packbits((want[:912].reshape(-1, 3).sum(1) >= 2).astype(np.uint8))
        pay = np.packbits((want[912:].reshape(-1, 7).sum(1) >= 4).astype(np.uint8))
        assert hdr.tobytes()[:4] == b"FTTG" and len(hdr) == 38 and len(pay) == plen
        nimg = 3
        H = np.tile(hdr, (nimg, 1)).copy(); P = np.tile(pay, (nimg, 1)).copy()
        bits = np.zeros((nimg, n), np.uint8)
        ctx.frame_expand_dev(nimg, H.ctypes.data, P.ctypes.data, plen, bits.ctypes.data)
        ctx.sync()
        assert all(np.array_equal(bits[i], want) for i in range(nimg))
        noisy = bits.copy(); noisy[:, ::5] ^= 1
        H2 = np.zeros_like(H); P2 = np.zeros_like(P)
        ctx.frame_majority_dev(nimg, noisy.ctypes.data, plen, H2.ctypes.data, P2.ctypes.data)
        ctx.sync()
        assert np.array_equal(H2, H) and np.array_equal(P2, P)
    ctx.close()


def test_context_reuse_across_geometries(emu, orc):
    PC.check_context_reuse(emu, orc, (2048, 160), [(2048, 130), (64, 64), (1500, 140), (40, 24), (700, 160), (2047, 129), (9, 5)])


def test_bit_index_address_order(emu, orc):
    PC.check_bit_index(emu, orc, 64, 48, 700)
    PC.check_bit_index(emu, orc, 100, 64, 500, jitter=0.05)


def test_audit64_is_the_reference_fft_bit_for_bit(emu, orc):
    PC.check_audit64_against_oracle(emu, orc, [(64, 64), (48, 40), (100, 30), (16, 1), (1, 8)])
    PC.check_product_against_audit64(emu, 200, 96)


@pytest.mark.parametrize("wh", [(40, 300), (2040, 130), (100, 64)])
def test_tile_resident_extraction_over_the_column_plans(emu, orc, wh):
    """Spectrum-free batched extraction (bins bucketed per tile, bits read in LDS by the last forward column step)
    against the spectrum + k_read path for a two-step column plan (PH = 512), the fused rows+columns plan
    (PW = 2048) and a direct plan whose half width is not a multiple of the 16-column tile."""
    from steganosaurus_amd.synth import cover_rgb
    w, h = wh
    nimg, n = 2, 200
    ph, pw = orc.next_pow2(h), orc.next_pow2(w)
    # rmax = 0.95: part of the walk lies in the mirror half (x > PW/2), stored conjugated at (PH-y, PW-x)
    bins = B.Walk(orc.subkeys(PC.PK)[0], ph, pw, rmin=0.05, rmax=0.95, lib=emu).next(n)
    if ph >= 4 * pw:        # tall grid: the annulus reaches beyond PW/2
        assert (bins["x"] > pw // 2).any() and (bins["x"] < pw // 2).any()
    sbins, idx = B.bins_sort(bins, lib=emu)
    imgs = np.stack([cover_rgb(w, h, 20 + i) for i in range(nimg)])
    res = []
    for mode, bl, index in (("3", sbins, idx), ("0", sbins, idx), ("3", bins, None), ("2", bins, None)):
        os.environ["TFFT_TILE_READ"] = mode
        try:
            ctx = B.Context(w, h, slots=nimg, lib=emu)
        finally:
            del os.environ["TFFT_TILE_READ"]
        if index is not None:
            ctx.set_bit_index(index)
        raw = np.full((nimg, n), 9, np.uint8)
        ctx.extract_batch_dev(nimg, imgs.ctypes.data, w, h, bl.ctypes.data, n, raw.ctypes.data)
        ctx.sync(); ctx.close()
        res.append(raw)
    for r in res[1:]:
        assert np.array_equal(r, res[0])
    assert set(np.unique(res[0])) <= {0, 1}


@pytest.mark.parametrize("case", [dict(w=40, h=24, n_bits=150), dict(w=40, h=300, n_bits=300, rmax=0.95), dict(w=2040, h=130, n_bits=300, nimg=1),
                                  dict(w=100, h=64, n_bits=200, center=True, sort=False),
                                  dict(w=64, h=64, n_bits=1500, rmax=0.95, nimg=3)])      # the last: buckets longer than the prefetch depth
def test_delta_embedding_over_the_column_plans(emu, orc, case):
    """stego = cover + IFFT(F' - F) with tiles built from the bucketed bins: direct, two-step (mirror half included) and fused plans"""
    lsb = 0.05 if case["w"] * case["h"] < 4096 else 0.01
    PC.check_delta_embedding(emu, orc, PC.HostBufs, lsb_frac=lsb, **case)


def test_tile_resident_extraction_buffers_grow_and_shrink(emu, orc):
    """One context, bin lists of growing and shrinking length (the bucket buffers are reallocated on demand) and
    two geometries in turn: always the bits of the spectrum + k_read path."""
    from steganosaurus_amd.synth import cover_rgb
    ctxs = []
    for mode in ("3", "0"):
        os.environ["TFFT_TILE_READ"] = mode
        try:
            ctxs.append(B.Context(200, 96, slots=2, lib=emu))
        finally:
            del os.environ["TFFT_TILE_READ"]
    for (w, h, n) in [(64, 48, 60), (200, 96, 4000), (64, 48, 900), (200, 80, 10)]:
        ph, pw = orc.next_pow2(h), orc.next_pow2(w)
        bins = B.Walk(orc.subkeys(PC.PK)[0], ph, pw, lib=emu).next(n)
        imgs = np.stack([cover_rgb(w, h, 3 + i) for i in range(2)])
        out = []
        for c in ctxs:
            raw = np.full((2, n), 5, np.uint8)
            c.extract_batch_dev(2, imgs.ctypes.data, w, h, bins.ctypes.data, n, raw.ctypes.data)
            c.sync()
            out.append(raw)
        assert np.array_equal(out[0], out[1]), (w, h, n)
    for c in ctxs:
        c.close()


def test_dc_removal_option(emu, orc):
    PC.check_dc_removal(emu, [(64, 64), (48, 40), (200, 96), (100, 300), (2040, 130)],
                        lambda img, center: orc.forward_rgb8(img, center=center)[0])
    # the whole embed / extract parity case (bins, stego pixels within 1 LSB, extracted bits) with the switch OFF
    # (the default-on path is what every other test of this file runs): both directions leave the constant in
    os.environ["TFFT_DC_BIAS"] = "0"
    try:
        PC.check_embed_extract(emu, orc, 64, 64, 600, dict(), spectrum_bars=False)
        PC.check_embed_extract(emu, orc, 128, 64, 500, dict(center=1), spectrum_bars=False)
    finally:
        del os.environ["TFFT_DC_BIAS"]


def test_cover_hash_matches_the_reference(emu, golden_dir):
    import ctypes as C
    host = C.CDLL(os.path.join(os.path.dirname(EMU_DIR), "..", "steganosaurus_amd", "libtfhost.so"))
    PC.check_cover_hash(emu, host, golden_dir, max_pixels=300 * 300)


def test_stream_batch_two_phase_extract(emu, orc):
    PC.check_stream_batch(emu, orc, PC.HostBufs, 256, 256, secrets=(8, 8, 20, 20, 8), slots=3, sort=True)
    PC.check_stream_batch(emu, orc, PC.HostBufs, 256, 128, secrets=(1,), slots=1, sort=False)


def test_batch_capacity_inside_the_median_pass(emu):
    small = ({}, {"TFFT_MEDIAN_FALLBACK": "1"}, {"TFFT_STATS_FUSED": "0"})      # (direct column plans: the in-kernel statistics never apply)
    PC.check_batch_capacity(emu, PC.HostBufs, 96, 64, nimg=2, envs=small)
    PC.check_batch_capacity(emu, PC.HostBufs, 40, 200, nimg=1, cases=((0.0, 1.5, 0.5),), envs=small)
    # fused 2048-wide plan: the statistics inside the last forward column step (COLS_STAT); a flat image among them
    PC.check_batch_capacity(emu, PC.HostBufs, 2040, 130, nimg=2, cases=((0.05, 0.45, 0.01), (0.05, 0.45, 1.0)), flat=True, envs=({"TFFT_STATS_TILE": "2"},))
