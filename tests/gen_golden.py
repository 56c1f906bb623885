#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE itself (oracle/_ref, built in place
from /root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle all && python tests/gen_golden.py

The reference ships no vectors for this path, so these
outputs -- inputs + expected outputs, never reference source -- are what pins
the oracle and the HIP path on the GPU box, where /root/reference does not exist.
Covers are regenerated from steganosaurus_amd.synth (seeded), so only expected
outputs are stored.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _checkers import Checker, Params, REF_CLI, bins_digest  # noqa: E402
from steganosaurus_amd.synth import cover_rgb, gradient_cover, secret_ascii, n_stream_bits  # noqa: E402

OUT = os.path.join(HERE, "golden")
PASS = "test123"
SALT = bytes(range(16))
ITERS = 1000

VARIANTS = {
    "default": dict(),
    "jitter": dict(jitter=0.05),
    "adaptive": dict(adaptive_alpha=1),
    "center": dict(center=1),
    "alpha_density": dict(alpha=0.3, density=0.5, rmin=0.1, rmax=0.6),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    R = Checker("ref")
    pk = hashlib.sha256(PASS.encode()).digest()            # S:1038
    kw, kr, kg, kb = R.subkeys(pk)
    kat = {"pass": PASS, "path_key": pk.hex(), "key_walk": kw.hex(), "key_r": kr.hex(), "key_g": kg.hex(),
           "key_b": kb.hex(), "ks_walk_first64": R.ks_bytes(kw, 64).tobytes().hex(),
           "opcodes_first64": R.ks_opcodes(kw, 64).tolist(), "walks": []}
    # --- walk KATs (square, tall, wide; short and long) ---------------------------------
    for (PH, PW, n) in [(64, 64, 300), (256, 256, 2480), (512, 512, 59152), (512, 1024, 20000), (1024, 512, 20000),
                        (2048, 2048, 59152), (4096, 4096, 59152), (8192, 8192, 59152), (2048, 2048, 231184)]:
        rc, bins, sk, ctr, start = R.walk(kw, PH, PW, n)
        kat["walks"].append({"PH": PH, "PW": PW, "n": n, "start": start.tolist(), "first8": bins[:8].tolist(),
                             "skipped": sk, "ks_ctr": ctr, "sha256": bins_digest(bins)})
        print("walk", PH, PW, n, bins_digest(bins)[:16], flush=True)
    for (rmin, rmax, dens, n) in [(0.1, 0.6, 0.5, 1500), (0.0, 1.0, 0.9, 1500), (0.2, 0.3, 0.25, 200)]:
        # n is kept well inside the annulus capacity: the reference spins forever once it is exhausted
        rc, bins, sk, ctr, start = R.walk(kw, 128, 256, n, rmin, rmax, dens)
        kat["walks"].append({"PH": 128, "PW": 256, "n": n, "rmin": rmin, "rmax": rmax, "density": dens,
                             "start": start.tolist(), "first8": bins[:8].tolist(), "skipped": sk, "ks_ctr": ctr,
                             "sha256": bins_digest(bins)})
    # --- FFT sign KAT (SURVEY finding 3) --------------------------------------------------
    d = np.zeros((4, 8), np.complex128); d[1, 1] = 1
    F = R.fft2d(d)
    kat["delta_4x8"] = {"F01": [F[0, 1].real, F[0, 1].imag], "F10": [F[1, 0].real, F[1, 0].imag]}
    # --- framing (host crypto) -------------------------------------------------------------
    kat["frames"] = []
    for slen in (1, 12, 54, 1024):
        sec = secret_ascii(slen, seed=slen)
        fb = R.frame_bits(PASS, SALT, ITERS, sec)
        assert len(fb) == n_stream_bits(slen)
        n, back = R.deframe_bits(PASS, ITERS, fb)
        assert back == sec
        kat["frames"].append({"secret": sec.decode(), "salt": SALT.hex(), "iters": ITERS,
                              "bits_packed": np.packbits(fb).tobytes().hex()})
    with open(os.path.join(OUT, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # --- full spectra, small ----------------------------------------------------------------
    for (W, H) in [(64, 64), (48, 40), (100, 30)]:
        img = cover_rgb(W, H, 0)
        for center in (0, 1):
            spec, med = R.forward_rgb8(img, center)
            cap, _ = R.capacity_rgb8(img, Params(center=center))
            np.savez_compressed(os.path.join(OUT, f"fft_{W}x{H}_c{center}.npz"), W=W, H=H, cover_index=0,
                                center=center, spec=spec, med=med, capacity=cap)
    # --- 512^2: sparse sample + norms -----------------------------------------------------
    for name, img in (("lcg", cover_rgb(512, 512, 0)), ("grad", gradient_cover(512, 512, 1))):
        spec, med = R.forward_rgb8(img)
        cap, _ = R.capacity_rgb8(img)
        rng = np.random.default_rng(99)
        pos = rng.integers(0, 512, size=(3, 4096, 2))
        vals = np.stack([spec[p][pos[p, :, 0], pos[p, :, 1]] for p in range(3)])
        np.savez_compressed(os.path.join(OUT, f"fft_512_{name}.npz"), pos=pos, vals=vals, med=med, capacity=cap,
                            l2=np.array([np.linalg.norm(spec[p]) for p in range(3)]),
                            col0=spec[:, :, 0], colN=spec[:, :, 256], row0=spec[:, 0, :])
    # --- embed / extract, small + config 1 ---------------------------------------------------
    rng = np.random.default_rng(5)
    for (W, H, n) in [(64, 64, 300), (48, 40, 300), (256, 256, 2480)]:
        img = cover_rgb(W, H, 0)
        bits = rng.integers(0, 2, n).astype(np.uint8)
        rec = {"W": W, "H": H, "cover_index": 0, "bits": bits}
        for vname, kw_ in VARIANTS.items():
            P = Params(**kw_)
            stego, spec, bins = R.embed_rgb8(img, pk, bits, P, want_spec=True, want_bins=True)
            raw = R.extract_bits(stego, pk, n, P)
            rec[f"{vname}_stego"] = stego
            rec[f"{vname}_bins"] = bins
            rec[f"{vname}_raw"] = raw
            if W <= 64:
                rec[f"{vname}_spec"] = spec
            print("embed", W, H, vname, "BER", float((raw != bits).mean()))
        np.savez_compressed(os.path.join(OUT, f"embed_{W}x{H}.npz"), **rec)
    # config 1: 512x512, 1 KB secret, defaults (BASELINE.json configs[0])
    for name, img in (("lcg", cover_rgb(512, 512, 0)), ("grad", gradient_cover(512, 512, 1))):
        sec = secret_ascii(1024, seed=1024)
        bits = R.frame_bits(PASS, SALT, ITERS, sec)
        stego, _, bins = R.embed_rgb8(img, pk, bits, Params(), want_bins=True)
        raw = R.extract_bits(stego, pk, len(bits), Params())
        n, back = R.deframe_bits(PASS, ITERS, raw)
        assert back == sec, "reference round trip failed at config 1"
        print("cfg1", name, "BER", float((raw != bits).mean()), "changed px", int((stego != img).sum()))
        np.savez_compressed(os.path.join(OUT, f"embed_512_{name}.npz"), bits=np.packbits(bits), n_bits=len(bits),
                            bins_sha256=bins_digest(bins), stego_diff=(stego.astype(np.int16) - img),
                            stego_sha256=hashlib.sha256(stego.tobytes()).hexdigest(), raw=np.packbits(raw))
    # --- non-pow2: the reference's own (erroneous) raw bits, SURVEY finding 1 ------------------
    for (W, H, n) in [(600, 400, 5000), (300, 500, 5000)]:
        img = cover_rgb(W, H, 0)
        bits = rng.integers(0, 2, n).astype(np.uint8)
        stego, _, bins = R.embed_rgb8(img, pk, bits, Params(), want_bins=True)
        raw = R.extract_bits(stego, pk, n, Params())
        print("nonpow2", W, H, "BER", float((raw != bits).mean()))
        np.savez_compressed(os.path.join(OUT, f"nonpow2_{W}x{H}.npz"), W=W, H=H, bits=np.packbits(bits), n_bits=n,
                            bins_sha256=bins_digest(bins), stego_diff=(stego.astype(np.int16) - img),
                            raw=np.packbits(raw))
    # --- CLI interop: a stego PNG made by the reference's real do_embed -----------------------
    with tempfile.TemporaryDirectory() as td:
        cover = os.path.join(td, "cover.png")
        R.png_write(cover, gradient_cover(256, 256, 3))
        stego = os.path.join(OUT, "cli_256_hello.png")
        subprocess.run([REF_CLI, "embed", "--in", cover, "--out", stego, "--secret", "Hello World!", "--pass", PASS,
                        "--pbkdf2_iter", str(ITERS)], check=True)
        out = subprocess.run([REF_CLI, "extract", "--in", stego, "--pass", PASS, "--pbkdf2_iter", str(ITERS)],
                             check=True, capture_output=True, text=True).stdout
        assert out == "Hello World!\n", out
        # harness composition == real CLI: raw bits of the CLI stego deframe to the secret
        raw = R.extract_bits(R.png_read(stego), pk, n_stream_bits(12), Params())
        n, back = R.deframe_bits(PASS, ITERS, raw)
        assert back == b"Hello World!"
        # and the other direction: harness-embedded image is accepted by the real do_extract
        img = gradient_cover(256, 256, 3)
        fb = R.frame_bits(PASS, SALT, ITERS, b"other way")
        st, _, _ = R.embed_rgb8(img, pk, fb, Params())
        p2 = os.path.join(td, "h.png"); R.png_write(p2, st)
        out = subprocess.run([REF_CLI, "extract", "--in", p2, "--pass", PASS, "--pbkdf2_iter", str(ITERS)],
                             check=True, capture_output=True, text=True).stdout
        assert out == "other way\n", out
    print("golden fixtures written to", OUT)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total bytes", tot)


if __name__ == "__main__":
    main()
