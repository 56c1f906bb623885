"""ctypes bindings of the two CHECKER libraries (test infrastructure only).

* ``oracle/liboracle.so``   -- this repo's CPU restatement (prefix ``orc_``)
* ``oracle/_ref/libtfref.so`` -- the reference TU compiled in place (prefix ``ref_``);
  present in the build container and, as a prebuilt file, on the GPU box.

Both export the same signatures, so ``Checker('orc')`` and ``Checker('ref')``
are interchangeable in the tests.  Nothing under ``steganosaurus_amd/`` imports
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libtfref.so")
REF_CLI = os.path.join(ORACLE_DIR, "_ref", "turtlefft")


class Params(C.Structure):
    """Field order of the reference's struct Params (S:375-381) that the signal path uses."""
    _fields_ = [("alpha", C.c_double), ("rmin", C.c_double), ("rmax", C.c_double), ("magmin", C.c_double),
                ("density", C.c_double), ("jitter", C.c_double), ("center", C.c_int), ("adaptive_alpha", C.c_int)]

    def __init__(self, alpha=0.50, rmin=0.05, rmax=0.45, magmin=0.01, density=0.7, jitter=0.0, center=0,
                 adaptive_alpha=0):
        super().__init__(alpha, rmin, rmax, magmin, density, jitter, int(center), int(adaptive_alpha))


def build_oracle():
    """(Re)build the checkers; the reference part only where /root/reference exists."""
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)


def have_ref():
    return os.path.exists(REF_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Checker:
    def __init__(self, kind="orc"):
        self.kind = kind
        path = ORACLE_SO if kind == "orc" else REF_SO
        if kind == "orc" and not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        L, px = self.lib, kind + "_"
        self._f = {}
        sig = {
            "next_pow2": (C.c_int, [C.c_int]),
            "sha256": (None, [C.c_char_p, C.c_size_t, C.c_void_p]),
            "subkeys": (None, [C.c_char_p, C.c_void_p]),
            "ks_bytes": (None, [C.c_char_p, C.c_size_t, C.c_void_p]),
            "ks_opcodes": (None, [C.c_char_p, C.c_size_t, C.c_void_p]),
            "fft1d": (None, [C.c_void_p, C.c_int, C.c_int]),
            "fft2d": (None, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
            "forward_rgb8": (None, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
            "capacity_rgb8": (C.c_uint64, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p]),
            "walk": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
            "embed_rgb8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_char_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
            "extract_bits": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_char_p, C.c_uint64,
                                       C.c_void_p]),
            "cover_hash": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
        }
        if kind == "ref":
            sig.update({
                "frame_bits": (C.c_uint64, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32,
                                            C.c_void_p, C.c_uint64]),
                "deframe_bits": (C.c_int64, [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_uint64]),
                "png_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
                "png_read": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
            })
        else:
            sig.update({
                "median_abs": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
                "read_bins": (None, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
            })
        for name, (res, args) in sig.items():
            fn = getattr(L, px + name)
            fn.restype, fn.argtypes = res, args
            self._f[name] = fn

    # -- small wrappers returning numpy -------------------------------------------------
    def next_pow2(self, v):
        return self._f["next_pow2"](v)

    def sha256(self, data: bytes) -> bytes:
        out = C.create_string_buffer(32)
        self._f["sha256"](data, len(data), out)
        return out.raw

    def subkeys(self, path_key: bytes):
        out = C.create_string_buffer(128)
        self._f["subkeys"](path_key, out)
        r = out.raw
        return r[:32], r[32:64], r[64:96], r[96:128]

    def ks_bytes(self, key, n):
        out = np.zeros(n, np.uint8)
        self._f["ks_bytes"](key, n, _p(out))
        return out

    def ks_opcodes(self, key, n):
        out = np.zeros(n, np.uint8)
        self._f["ks_opcodes"](key, n, _p(out))
        return out

    def fft1d(self, a, inverse=False):
        a = np.ascontiguousarray(a, np.complex128).copy()
        self._f["fft1d"](_p(a), a.shape[0], int(inverse))
        return a

    def fft2d(self, a, inverse=False):
        a = np.ascontiguousarray(a, np.complex128).copy()
        self._f["fft2d"](_p(a), a.shape[0], a.shape[1], int(inverse))
        return a

    def forward_rgb8(self, rgb, center=False, want_spec=True):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        PH, PW = self.next_pow2(H), self.next_pow2(W)
        spec = np.zeros((3, PH, PW), np.complex128) if want_spec else None
        med = np.zeros(3, np.float64)
        self._f["forward_rgb8"](_p(rgb), W, H, int(center), _p(spec), _p(med))
        return spec, med

    def capacity_rgb8(self, rgb, params=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        med = np.zeros(3, np.float64)
        p = params or Params()
        cap = self._f["capacity_rgb8"](_p(rgb), W, H, C.byref(p), _p(med))
        return int(cap), med

    def walk(self, key_walk, PH, PW, n, rmin=0.05, rmax=0.45, density=0.7):
        bins = np.zeros((n, 3), np.int32)
        sk, ctr = C.c_uint64(0), C.c_uint32(0)
        start = np.zeros(3, np.int32)
        rc = self._f["walk"](key_walk, PH, PW, rmin, rmax, density, n, _p(bins), C.byref(sk), C.byref(ctr), _p(start))
        return rc, bins, sk.value, ctr.value, start

    def embed_rgb8(self, rgb, path_key, bits, params=None, want_spec=False, want_bins=False):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        bits = np.ascontiguousarray(bits, np.uint8)
        H, W = rgb.shape[:2]
        PH, PW = self.next_pow2(H), self.next_pow2(W)
        out = np.zeros_like(rgb)
        spec = np.zeros((3, PH, PW), np.complex128) if want_spec else None
        bins = np.zeros((len(bits), 3), np.int32) if want_bins else None
        p = params or Params()
        rc = self._f["embed_rgb8"](_p(rgb), W, H, C.byref(p), path_key, _p(bits), len(bits), _p(out), _p(spec), _p(bins))
        if rc != 0:
            raise RuntimeError("checker embed failed rc=%d" % rc)
        return out, spec, bins

    def extract_bits(self, rgb, path_key, n_bits, params=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        out = np.zeros(n_bits, np.uint8)
        p = params or Params()
        rc = self._f["extract_bits"](_p(rgb), W, H, C.byref(p), path_key, n_bits, _p(out))
        if rc != 0:
            raise RuntimeError("checker extract failed rc=%d" % rc)
        return out

    def cover_hash(self, rgb, center=False):
        """compute_cover_hash S:415-444 -> (region, hash32, magnitudes[3*region^2], quantised bytes)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        H, W = rgb.shape[:2]
        h = np.zeros(32, np.uint8); mags = np.zeros(192, np.float64); q = np.zeros(192, np.uint8)
        region = self._f["cover_hash"](_p(rgb), W, H, int(center), _p(h), _p(mags), _p(q))
        n = 3 * region * region
        return region, h.tobytes(), mags[:n].copy(), q[:n].copy()

    # -- oracle only -------------------------------------------------------------------
    def median_abs(self, plane):
        plane = np.ascontiguousarray(plane, np.complex128)
        return self._f["median_abs"](_p(plane), plane.shape[0], plane.shape[1])

    def read_bins(self, spec, bins, alpha=0.5, jitter=None, adaptive=False, med=None):
        spec = np.ascontiguousarray(spec, np.complex128)
        bins = np.ascontiguousarray(bins, np.int32)
        out = np.zeros(len(bins), np.uint8)
        jit = np.ascontiguousarray(jitter, np.float64) if jitter is not None else None
        m = np.ascontiguousarray(med, np.float64) if med is not None else None
        self._f["read_bins"](_p(spec), spec.shape[1], spec.shape[2], _p(bins), _p(jit), len(bins), alpha,
                             int(adaptive), _p(m), _p(out))
        return out

    # -- reference only ----------------------------------------------------------------
    def frame_bits(self, password: str, salt: bytes, iters: int, secret: bytes):
        cap = 38 * 24 + (len(secret) + 16) * 56
        out = np.zeros(cap, np.uint8)
        n = self._f["frame_bits"](password.encode(), salt, iters, secret, len(secret), _p(out), cap)
        assert n == cap
        return out

    def deframe_bits(self, password: str, iters: int, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(max(1, len(bits) // 56), np.uint8)
        n = self._f["deframe_bits"](password.encode(), iters, _p(bits), len(bits), _p(out), len(out))
        return n, bytes(out[:max(0, n)])

    def png_write(self, path, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert self._f["png_write"](path.encode(), _p(rgb), rgb.shape[1], rgb.shape[0]) == 0

    def png_read(self, path):
        W, H = C.c_int(0), C.c_int(0)
        assert self._f["png_read"](path.encode(), None, 0, C.byref(W), C.byref(H)) == 0
        out = np.zeros((H.value, W.value, 3), np.uint8)
        assert self._f["png_read"](path.encode(), _p(out), out.size, C.byref(W), C.byref(H)) == 0
        return out


def bins_digest(bins) -> str:
    """SHA-256 over [u8 plane | u32le y | u32le x]* -- the serialisation SURVEY.md 8(c) used."""
    import hashlib
    b = np.ascontiguousarray(bins, np.int32)
    rec = np.zeros(len(b), dtype=[("p", "u1"), ("y", "<u4"), ("x", "<u4")])
    rec["p"], rec["y"], rec["x"] = b[:, 0], b[:, 1], b[:, 2]
    return hashlib.sha256(rec.tobytes()).hexdigest()
