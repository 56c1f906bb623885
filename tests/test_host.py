"""CPU tests of everything around the kernels: the C-ABI surface, the host walk, the CLI's
crypto/framing/PNG (against RFC vectors, Python's stdlib and reference-made goldens), the
no-GPU failure mode, and the 2-rank sharding logic (gloo)."""
import ctypes as C
import hashlib
import hmac
import json
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

from _checkers import bins_digest
from steganosaurus_amd import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "steganosaurus_amd")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(PKG, "libtfhost.so"))
    lib.tfh_frame_bits.restype = C.c_uint64
    lib.tfh_deframe_bits.restype = C.c_int64
    return lib


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "kat.json")) as f:
        return json.load(f)


# ------------------------------------------------------------------ C ABI surface
def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "turtlefft_hip.h")).read()
    declared = set(re.findall(r"\b(tfft_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tfft_status"}
    assert len(declared) >= 30
    lib = B.load()                                   # binds every SYMBOLS entry or raises
    exported = subprocess.run(["nm", "-D", "--defined-only", B.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, exported), "header declares %s but the library does not export it" % name
        assert name in B.SYMBOLS, "binding.py does not bind %s" % name
    assert set(B.SYMBOLS) <= declared
    assert lib.tfft_abi_version() == 1
    assert b"no CPU fallback" in lib.tfft_strerror(-2)
    # the PNG pipeline helper (include/turtlefft_pipe.h -> libtfpipe.so)
    hdr = open(os.path.join(ROOT, "include", "turtlefft_pipe.h")).read()
    declared = set(re.findall(r"\b(tfp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {"tfp_embed_png_batch", "tfp_extract_png_batch"}
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, "libtfpipe.so")], capture_output=True, text=True, check=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, exported), name
    pipe = B.load_pipe()
    assert pipe.tfp_embed_png_batch and pipe.tfp_extract_png_batch


def test_no_gpu_means_loud_failure():
    """Without a usable gfx950 device context creation fails; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(B.TfftError) as e:
        B.Context(64, 64)
    assert e.value.status == -2


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "libtfref" not in txt and "_checkers" not in txt, f
                assert "tfft_emu" not in txt, f


# ------------------------------------------------------------------ host walk through the product library
def test_walk_goldens_product_library(kat):
    kw = bytes.fromhex(kat["key_walk"])
    for w in kat["walks"]:
        wk = B.Walk(kw, w["PH"], w["PW"], w.get("rmin", 0.05), w.get("rmax", 0.45), w.get("density", 0.7))
        assert list(wk.start()) == w["start"]
        bins = wk.next(w["n"])
        t = B.bins_to_triples(bins)
        assert t[:8].tolist() == w["first8"]
        assert bins_digest(t) == w["sha256"], (w["PH"], w["PW"], w["n"])
        assert (wk.skipped, wk.ks_blocks()) == (w["skipped"], w["ks_ctr"])
        wk.close()


def test_walk_exhaustion_and_bad_args():
    kw = bytes(32)
    wk = B.Walk(kw, 32, 32)
    with pytest.raises(B.TfftError) as e:
        wk.next(5000)
    assert e.value.status == -7                       # TFFT_E_EXHAUSTED, where the reference spins forever
    wk = B.Walk(kw, 32, 32, density=1.0)              # (uint8_t)256 == 0: never hits
    with pytest.raises(B.TfftError):
        wk.next(1)
    with pytest.raises(B.TfftError):
        B.Walk(kw, 0, 32)
    # PH == 1 / PW == 1: every bin is on an axis
    wk = B.Walk(kw, 1, 64)
    with pytest.raises(B.TfftError):
        wk.next(1)


def test_walk_jitter_matches_oracle(orc, kat):
    pk = bytes.fromhex(kat["path_key"])
    sub = orc.subkeys(pk)
    wk = B.Walk(sub[0], 64, 64)
    bins = wk.next(200)
    j = B.walk_jitter(sub[1] + sub[2] + sub[3], bins, 0.05)
    # oracle: per-plane keystreams, two bytes per bin, high byte first (S:690-694)
    ks = [orc.ks_bytes(sub[1 + p], 2 * 200) for p in range(3)]
    pos = [0, 0, 0]
    for i, b in enumerate(bins):
        p = int(b["plane"]); hi, lo = int(ks[p][pos[p]]), int(ks[p][pos[p] + 1]); pos[p] += 2
        r = (hi << 8) | lo
        r = r - 65536 if r >= 32768 else r
        assert abs(j[i] - np.float32(r / 32768.0 * 0.05)) < 1e-9


# ------------------------------------------------------------------ host crypto
def test_sha_hmac_pbkdf2_hkdf_against_stdlib(host):
    rng = np.random.default_rng(0)
    out = C.create_string_buffer(64)
    for n in (0, 1, 55, 56, 63, 64, 65, 1000):
        msg = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        host.tfh_sha256(msg, C.c_size_t(n), out)
        assert out.raw[:32] == hashlib.sha256(msg).digest()
        for kl in (1, 32, 64, 65, 200):
            key = rng.integers(0, 256, kl, dtype=np.uint8).tobytes()
            host.tfh_hmac_sha256(key, C.c_size_t(kl), msg, C.c_size_t(n), out)
            assert out.raw[:32] == hmac.new(key, msg, hashlib.sha256).digest()
    for (pw, salt, it, dk) in [(b"passwd", b"salt", 1, 64), (b"Password", b"NaCl", 80000, 64), (b"test123", bytes(range(16)), 1000, 32)]:
        host.tfh_pbkdf2(pw, C.c_size_t(len(pw)), salt, C.c_size_t(len(salt)), it, out, C.c_size_t(dk))
        assert out.raw[:dk] == hashlib.pbkdf2_hmac("sha256", pw, salt, it, dk)
    # RFC 7914 section 11, first vector
    host.tfh_pbkdf2(b"passwd", C.c_size_t(6), b"salt", C.c_size_t(4), 1, out, C.c_size_t(64))
    assert out.raw[:8].hex() == "55ac046e56e3089f"
    # RFC 5869 test case 1
    ikm, salt, info = bytes([0x0b] * 22), bytes(range(13)), bytes(range(0xf0, 0xfa))
    prk = C.create_string_buffer(32)
    host.tfh_hkdf_extract(salt, C.c_size_t(13), ikm, C.c_size_t(22), prk)
    assert prk.raw.hex() == "077709362c2e32df0ddc3f0dc47bba6390b6c73bb50f9c3122ec844ad7c2b3e5"
    okm = C.create_string_buffer(42)
    host.tfh_hkdf_expand(prk, info, C.c_size_t(10), okm, C.c_size_t(42))
    assert okm.raw.hex() == "3cb25f25faacd57a90434f64d0362f2a2d2d0a90cf1a5a4c5db02d56ecc4c5bf34007208d5b887185865"


def test_chacha20_poly1305_rfc8439(host):
    key = bytes(range(0x80, 0xa0))
    nonce = bytes.fromhex("070000004041424344454647")
    aad = bytes.fromhex("50515253c0c1c2c3c4c5c6c7")
    pt = (b"Ladies and Gentlemen of the class of '99: If I could offer you only one tip for the future, "
          b"sunscreen would be it.")
    ct = C.create_string_buffer(len(pt)); tag = C.create_string_buffer(16)
    host.tfh_aead_seal(key, nonce, aad, C.c_size_t(len(aad)), pt, C.c_size_t(len(pt)), ct, tag)
    assert tag.raw.hex() == "1ae10b594f09e26a7e902ecbd0600691"
    assert ct.raw[:16].hex() == "d31a8d34648e60db7b86afbc53ef7ec2"
    back = C.create_string_buffer(len(pt))
    assert host.tfh_aead_open(key, nonce, aad, C.c_size_t(len(aad)), ct, C.c_size_t(len(pt)), tag, back) == 1
    assert back.raw == pt
    bad = bytearray(tag.raw); bad[0] ^= 1
    assert host.tfh_aead_open(key, nonce, aad, C.c_size_t(len(aad)), ct, C.c_size_t(len(pt)), bytes(bad), back) == 0
    # empty plaintext / empty aad
    host.tfh_aead_seal(key, nonce, None, C.c_size_t(0), None, C.c_size_t(0), ct, tag)
    assert host.tfh_aead_open(key, nonce, None, C.c_size_t(0), ct, C.c_size_t(0), tag, back) == 1


def test_framing_matches_reference_goldens(host, kat):
    """Rep-3(header) || Rep-7(ct||tag) bit for bit as the reference's do_embed produced it (fixed salt)."""
    for fr in kat["frames"]:
        secret = fr["secret"].encode()
        want = np.unpackbits(np.frombuffer(bytes.fromhex(fr["bits_packed"]), np.uint8))
        n = 38 * 24 + (len(secret) + 16) * 56
        out = np.zeros(n, np.uint8)
        got_n = host.tfh_frame_bits(kat["pass"].encode(), bytes.fromhex(fr["salt"]), fr["iters"], secret, len(secret),
                                    out.ctypes.data_as(C.c_void_p), C.c_uint64(n))
        assert got_n == n and np.array_equal(out, want[:n])
        back = C.create_string_buffer(len(secret) + 1)
        r = host.tfh_deframe_bits(kat["pass"].encode(), fr["iters"], out.ctypes.data_as(C.c_void_p), C.c_uint64(n), back,
                                  C.c_uint64(len(secret)))
        assert r == len(secret) and back.raw[:r] == secret
        # a few flipped bits are absorbed by the repetition codes
        noisy = out.copy(); noisy[::11] ^= 1
        r = host.tfh_deframe_bits(kat["pass"].encode(), fr["iters"], noisy.ctypes.data_as(C.c_void_p), C.c_uint64(n), back,
                                  C.c_uint64(len(secret)))
        assert r == len(secret) and back.raw[:r] == secret
        assert host.tfh_deframe_bits(b"wrong", fr["iters"], out.ctypes.data_as(C.c_void_p), C.c_uint64(n), back,
                                     C.c_uint64(len(secret))) == -4


def test_subkeys_match_goldens(host, kat):
    pk = C.create_string_buffer(32); sub = C.create_string_buffer(128)
    host.tfh_turtle_subkeys(kat["pass"].encode(), C.c_size_t(len(kat["pass"])), pk, sub)
    assert pk.raw.hex() == kat["path_key"]
    assert sub.raw.hex() == kat["key_walk"] + kat["key_r"] + kat["key_g"] + kat["key_b"]


# ------------------------------------------------------------------ PNG codec
def _png(w, h, ctype, depth, rows, plte=None, interlace=0):
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    raw = b"".join(b"\x00" + r for r in rows)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if plte:
        out += chunk(b"PLTE", plte)
    return out + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def _read(host, path):
    w, h = C.c_int(), C.c_int()
    assert host.tfh_image_read(path.encode(), None, C.c_uint64(0), C.byref(w), C.byref(h)) == 0
    out = np.zeros((h.value, w.value, 3), np.uint8)
    assert host.tfh_image_read(path.encode(), out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size), C.byref(w), C.byref(h)) == 0
    return out


def test_png_roundtrip_and_variants(host, tmp_path, golden_dir):
    rng = np.random.default_rng(3)
    for (w, h) in [(1, 1), (7, 5), (64, 64), (257, 33)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"rt_{w}x{h}.png")
        assert host.tfh_png_write(p.encode(), img.ctypes.data_as(C.c_void_p), w, h) == 0
        assert np.array_equal(_read(host, p), img)
        # an independent reader (zlib + filter undo in numpy terms) agrees on the container
        raw = open(p, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n" and b"IEND" in raw
    g = rng.integers(0, 256, (6, 9), dtype=np.uint8)
    cases = {
        "gray8": (_png(9, 6, 0, 8, [bytes(r) for r in g]), np.repeat(g[:, :, None], 3, 2)),
        "ga8": (_png(9, 6, 4, 8, [bytes(np.stack([r, 255 - r], 1).ravel()) for r in g]), np.repeat(g[:, :, None], 3, 2)),
        "rgba8": (_png(3, 6, 6, 8, [bytes(np.concatenate([r.reshape(3, 3), [[9], [9], [9]]], 1).ravel().astype(np.uint8)) for r in g]),
                  g.reshape(6, 3, 3)),
        "rgb16": (_png(3, 6, 2, 16, [bytes(np.stack([r, r ^ 0x5a], 1).ravel()) for r in g]), g.reshape(6, 3, 3)),
        "pal8": (_png(9, 6, 3, 8, [bytes(r % 4) for r in g], plte=bytes(range(12))),
                 np.arange(12, dtype=np.uint8).reshape(4, 3)[g % 4]),
    }
    for name, (data, want) in cases.items():
        p = str(tmp_path / (name + ".png"))
        open(p, "wb").write(data)
        assert np.array_equal(_read(host, p), want), name
    # the reference CLI's own PNG (stb writer) decodes
    ref_png = os.path.join(golden_dir, "cli_256_hello.png")
    assert _read(host, ref_png).shape == (256, 256, 3)
    # PPM
    p = str(tmp_path / "a.ppm")
    img = rng.integers(0, 256, (4, 5, 3), dtype=np.uint8)
    open(p, "wb").write(b"P6\n# c\n5 4\n255\n" + img.tobytes())
    assert np.array_equal(_read(host, p), img)
    w, h = C.c_int(), C.c_int()
    assert host.tfh_image_read(b"/nonexistent.png", None, C.c_uint64(0), C.byref(w), C.byref(h)) == -1


def test_png_matches_reference_decoder(host, ref, golden_dir, tmp_path):
    p = os.path.join(golden_dir, "cli_256_hello.png")
    assert np.array_equal(_read(host, p), ref.png_read(p))
    img = np.random.default_rng(5).integers(0, 256, (37, 41, 3), dtype=np.uint8)
    q = str(tmp_path / "mine.png")
    assert host.tfh_png_write(q.encode(), img.ctypes.data_as(C.c_void_p), 41, 37) == 0
    assert np.array_equal(ref.png_read(q), img)          # stb reads what we write


# ------------------------------------------------------------------ CLI without a GPU
CLI = os.path.join(PKG, "turtlefft")


def test_cli_argument_handling(tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage:")
    r = subprocess.run([CLI, "embed", "--frobnicate", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Unknown arg: --frobnicate\nUsage:")
    r = subprocess.run([CLI, "embed", "--in", "x.png", "--out", "y.png", "--secret", "s"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr                 # neither --pass nor --key
    r = subprocess.run([CLI, "extract", "--in", str(tmp_path / "missing.png"), "--pass", "p"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == "Failed to load %s\n" % (tmp_path / "missing.png")
    r = subprocess.run([CLI, "embed", "--in", "a", "--out", "b", "--secret", "c", "--pass", "d", "--alpha"], capture_output=True, text=True)
    assert r.returncode == 1                                           # missing value


def test_cli_gen_key(tmp_path):
    r = subprocess.run([CLI, "gen-key"], capture_output=True, text=True, check=True)
    m = re.search(r"Base64: (\S+)\n  Fingerprint: ([0-9a-f]{16})\n", r.stdout)
    import base64
    key = base64.b64decode(m.group(1))
    assert len(key) == 32 and hashlib.sha256(key).hexdigest()[:16] == m.group(2)
    kf = tmp_path / "k.txt"
    r = subprocess.run([CLI, "gen-key", "--key-out", str(kf)], capture_output=True, text=True, check=True)
    assert "Exported (unencrypted) to: %s" % kf in r.stdout
    assert base64.b64decode(kf.read_text().strip()) == base64.b64decode(re.search(r"Base64: (\S+)", r.stdout).group(1))
    wf = tmp_path / "w.txt"
    r = subprocess.run([CLI, "gen-key", "--key-out", str(wf), "--wrap-pass", "pw", "--pbkdf2_iter", "1000"],
                       capture_output=True, text=True, check=True)
    blob = base64.b64decode(wf.read_text().strip())
    assert len(blob) == 80 and blob[:4] == b"TFKW"            # the reference's 80-byte wrapped format (S:594-596)
    derived = hashlib.pbkdf2_hmac("sha256", b"pw", blob[4:20], 1000, 44)
    assert blob[20:32] == derived[32:44]


# ------------------------------------------------------------------ multi-rank logic (gloo, world_size 2)
def _rank_main(rank, world, port, tmpdir, kat_key_walk):
    import torch
    import torch.distributed as dist
    from steganosaurus_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_bits = 5000
    bins = torch.zeros((n_bits, 8), dtype=torch.uint8)
    index = torch.zeros(n_bits, dtype=torch.int64)
    if rank == 0:
        wk = B.Walk(bytes.fromhex(kat_key_walk), 256, 256)
        sb, idx = B.bins_sort(wk.next(n_bits))              # address order + the bit index that goes with it
        bins.copy_(torch.from_numpy(sb.view(np.uint8).reshape(-1, 8).copy()))
        index.copy_(torch.from_numpy(idx.astype(np.int64)))
    D.broadcast_bins(bins, 0, bit_index=index)
    assert sorted(index.tolist()) == list(range(n_bits))
    lo, hi = D.shard(7, rank, world)
    raw = torch.full((4, 16), rank, dtype=torch.uint8)
    got = D.gather_bits(raw, 0)
    t = D.max_over_ranks(1.0 + rank)
    np.save(os.path.join(tmpdir, "r%d.npy" % rank), np.array([hashlib.sha256(bins.numpy().tobytes() + index.numpy().tobytes()).hexdigest(), lo, hi, t,
                                                              -1 if got is None else len(got)], dtype=object), allow_pickle=True)
    dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast(tmp_path, kat):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_rank_main, args=(2, port, str(tmp_path), kat["key_walk"]), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npy", allow_pickle=True); r1 = np.load(tmp_path / "r1.npy", allow_pickle=True)
    assert r0[0] == r1[0]                                   # both ranks hold the same bin list
    assert (r0[1], r0[2], r1[1], r1[2]) == (0, 4, 4, 7)    # 7 images over 2 ranks
    assert r0[3] == r1[3] == 2.0                            # MAX over ranks
    assert r0[4] == 2 and r1[4] == -1


def test_shard_covers_everything():
    from steganosaurus_amd.dist import shard
    for n in (0, 1, 7, 256):
        for ws in (1, 2, 3, 8):
            spans = [shard(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))


def test_wrapped_key_from_reference_cli_unwraps(host, tmp_path):
    """The reference's gen-key wraps with its library AEAD, whose Poly1305 tag assembly has the same
    non-standard carry handling as the in-TU copy: only the compatible variant opens it."""
    from _checkers import REF_CLI
    if not os.path.exists(REF_CLI):
        pytest.skip("reference CLI not built")
    import base64
    kf = tmp_path / "w.txt"
    r = subprocess.run([REF_CLI, "gen-key", "--key-out", str(kf), "--wrap-pass", "pw", "--pbkdf2_iter", "1000"],
                       capture_output=True, text=True, check=True)
    key = base64.b64decode(re.search(r"Base64: (\S+)", r.stdout).group(1))
    blob = base64.b64decode(kf.read_text().strip())
    assert len(blob) == 80 and blob[:4] == b"TFKW"
    derived = hashlib.pbkdf2_hmac("sha256", b"pw", blob[4:20], 1000, 44)
    out = C.create_string_buffer(32)
    assert host.tfh_aead_open_turtle(derived[:32], blob[20:32], None, C.c_size_t(0), blob[32:64], C.c_size_t(32), blob[64:80], out) == 1
    assert out.raw == key
    assert host.tfh_aead_open(derived[:32], blob[20:32], None, C.c_size_t(0), blob[32:64], C.c_size_t(32), blob[64:80], out) == 0


def test_capacity_threshold_transform_matches_reference_compare():
    """k_capacity tests !(m2 < T2) on |F|^2 instead of the reference's !((double)|F| < thr) (S:1004):
    T2 = tfft_internal_mag2_threshold(thr) must make the two identical for every float, in particular
    for the neighbours of the boundary."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "steganosaurus_amd", "libturtlefft_hip.so"))
    f = lib.tfft_internal_mag2_threshold
    f.restype = C.c_float
    f.argtypes = [C.c_double]
    rng = np.random.default_rng(7)
    thrs = np.concatenate([10.0 ** rng.uniform(-6, 9, 300), np.float32(10.0 ** rng.uniform(-3, 6, 100)).astype(np.float64),
                           [0.0, -1.0, 1e-30, 3.0e38, 1e39, np.inf, np.nan, 1.0, 4.0, 2.0 ** -126]])
    for thr in thrs:
        t2 = np.float32(f(float(thr)))
        if np.isfinite(t2) and t2 > 0:
            around = np.float32(t2) * np.ones(129, np.float32)
            bits = around.view(np.uint32).astype(np.int64) + np.arange(-64, 65)
            m2 = np.clip(bits, 0, 0x7F7FFFFF).astype(np.uint32).view(np.float32)
        else:
            m2 = np.array([0.0, 1e-45, 1.0, 3.4e38, np.inf], np.float32)
        m2 = np.concatenate([m2, np.float32(10.0 ** rng.uniform(-12, 18, 64))]).astype(np.float32)
        with np.errstate(invalid="ignore"):
            ref = ~(np.sqrt(m2).astype(np.float64) < thr)        # np.sqrt on float32 is correctly rounded, like sqrtf
            got = ~(m2 < t2)
        assert (ref == got).all(), (thr, t2, m2[ref != got][:4])


def test_bins_sort_orders_by_address_and_returns_the_walk_positions():
    """tfft_bins_sort: strictly increasing (plane, y, x) keys (the walk never repeats a bin), sorted[i] ==
    walk[bit_index[i]], bit_index a permutation; degenerate sizes."""
    import steganosaurus_amd as S
    key = hashlib.sha256(b"sort").digest()
    for (ph, pw, n) in ((256, 256, 2480), (512, 2048, 30000), (64, 64, 1), (64, 64, 0)):
        bins = S.Walk(key, ph, pw).next(n)
        sb, idx = S.bins_sort(bins)
        assert sb.dtype == S.BIN_DTYPE and len(sb) == n and idx.dtype == np.uint32
        assert np.array_equal(sb, bins[idx]) and np.array_equal(np.sort(idx), np.arange(n))
        k = (sb["plane"].astype(np.int64) << 32) | (sb["y"].astype(np.int64) << 16) | sb["x"]
        assert (np.diff(k) > 0).all()
    # duplicates keep their walk order (stable): a synthetic list
    b = np.zeros(6, S.BIN_DTYPE); b["x"] = [5, 3, 5, 3, 1, 5]; b["y"] = 7; b["plane"] = [1, 0, 1, 0, 2, 0]
    sb, idx = S.bins_sort(b)
    assert idx.tolist() == [1, 3, 5, 0, 2, 4]


def test_cover_hash_quantiser_and_path_key(host, golden_dir):
    """The CLI's half of compute_cover_hash (S:433-443) and the path key it feeds (S:1020-1033): quantiser + SHA-256 over
    the reference's own magnitudes give the reference's hash; path_key = SHA256(secret || cover_hash)."""
    import parity_cases as PC
    for c in PC.load_cover_hash_cases(golden_dir):
        assert PC.host_cover_hash(host, np.array(c["mags"])).hex() == c["hash"], (c["w"], c["h"])
        q = np.minimum(7, np.maximum(0, np.floor(np.log(1.0 + np.array(c["mags"])) / 2.0))).astype(np.uint8)
        assert q.tolist() == c["q"] and hashlib.sha256(q.tobytes()).hexdigest() == c["hash"]
    ch = bytes.fromhex(PC.load_cover_hash_cases(golden_dir)[0]["hash"])
    out = C.create_string_buffer(32)
    host.tfh_path_key(b"test123", C.c_size_t(7), ch, out)
    assert out.raw == hashlib.sha256(b"test123" + ch).digest()
    host.tfh_path_key(b"test123", C.c_size_t(7), None, out)
    assert out.raw == hashlib.sha256(b"test123").digest()
