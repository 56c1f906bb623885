"""ctypes binding of the C ABI in include/turtlefft_hip.h.

This is plumbing for tests and bench.py: the product is the shared library
``steganosaurus_amd/libturtlefft_hip.so`` (hand-written HIP for gfx950) and the
C++ ``turtlefft`` CLI that links it.  There is no CPU fallback here: if the
library is missing, or no MI355X is visible, loading / context creation raises.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# TFFT_LIB: an A/B build of the device library (csrc/Makefile `variant`); measurements only
LIB_PATH = os.environ.get("TFFT_LIB") or os.path.join(_PKG, "libturtlefft_hip.so")

TFFT_OK = 0
STATUS = {0: "TFFT_OK", -1: "TFFT_E_INVALID", -2: "TFFT_E_NO_DEVICE", -3: "TFFT_E_TOO_LARGE", -4: "TFFT_E_NOMEM",
          -5: "TFFT_E_HIP", -6: "TFFT_E_STATE", -7: "TFFT_E_EXHAUSTED", -8: "TFFT_E_BIN_RANGE"}

# numpy view of struct tfft_bin {uint16 x; uint16 y; uint8 plane; uint8 rsv[3];}
BIN_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("plane", "u1"), ("rsv", "u1", (3,))])
assert BIN_DTYPE.itemsize == 8

# every symbol include/turtlefft_hip.h declares: (restype, argtypes)
_vp, _i, _d, _u64 = C.c_void_p, C.c_int, C.c_double, C.c_uint64
_pi = C.POINTER(C.c_int)
SYMBOLS = {
    "tfft_abi_version": (_i, []),
    "tfft_strerror": (C.c_char_p, [_i]),
    "tfft_create": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "tfft_destroy": (_i, [_vp]),
    "tfft_set_stream": (_i, [_vp, _vp]),
    "tfft_sync": (_i, [_vp]),
    "tfft_last_hip_error": (_i, [_vp]),
    "tfft_device_bytes": (C.c_size_t, [_vp]),
    "tfft_forward_rgb8": (_i, [_vp, _i, _vp, _i, _i, _i, _pi, _pi]),
    "tfft_forward_rgb8_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _pi, _pi]),
    "tfft_medians": (_i, [_vp, _i, _vp]),
    "tfft_median_path": (_i, [_vp, _i, _vp]),
    "tfft_capacity": (_i, [_vp, _i, _d, _d, _vp, C.POINTER(_u64)]),
    "tfft_lowfreq_mag": (_i, [_vp, _i, _i, _vp]),
    "tfft_embed_bins": (_i, [_vp, _i, _vp, _vp, _vp, _u64, _d, _i, _vp]),
    "tfft_exact_info": (_i, [_vp, _vp]),
    "tfft_embed_bins_dev": (_i, [_vp, _i, _vp, _vp, _vp, _u64, _d, _i, _vp]),
    "tfft_read_bins": (_i, [_vp, _i, _vp, _vp, _u64, _d, _i, _vp, _vp]),
    "tfft_read_bins_dev": (_i, [_vp, _i, _vp, _vp, _u64, _d, _i, _vp, _vp]),
    "tfft_inverse_rgb8": (_i, [_vp, _i, _vp]),
    "tfft_inverse_rgb8_dev": (_i, [_vp, _i, _vp]),
    "tfft_download_spectrum": (_i, [_vp, _i, _vp]),
    "tfft_embed_batch_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _u64, _d, _d, _d, _d, _vp, _vp]),
    "tfft_extract_batch_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _d, _vp]),
    "tfft_frame_expand_dev": (_i, [_vp, _i, _vp, _vp, _u64, _vp]),
    "tfft_frame_majority_dev": (_i, [_vp, _i, _vp, _u64, _vp, _vp]),
    "tfft_embed_batch": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _u64, _d, _d, _d, _d, _vp, _vp]),
    "tfft_extract_batch": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _d, _vp]),
    "tfft_host_alloc": (_vp, [C.c_size_t]),
    "tfft_host_free": (None, [_vp]),
    "tfft_plan_info": (_i, [_vp, _i, _i, _i, _pi]),
    "tfft_embed_stream_batch": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _vp, _vp, _u64, _d, _d, _d, _d, _vp, _vp]),
    "tfft_extract_stream_batch": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _d, _vp, _vp, _u64, _vp, _vp]),
    "tfft_embed_stream_batch_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _vp, _vp, _u64, _d, _d, _d, _d, _vp, _vp]),
    "tfft_extract_stream_batch_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _u64, _d, _vp, _vp, _u64, _vp, _vp]),
    "tfft_walk_create": (_i, [C.c_char_p, _i, _i, _d, _d, _d, C.POINTER(_vp)]),
    "tfft_walk_next": (_i, [_vp, _u64, _vp, C.POINTER(_u64)]),
    "tfft_walk_start": (_i, [_vp, _pi, _pi, _pi]),
    "tfft_walk_ks_blocks": (C.c_uint32, [_vp]),
    "tfft_walk_destroy": (_i, [_vp]),
    "tfft_walk_jitter": (_i, [C.c_char_p, _vp, _u64, _d, _vp]),
    "tfft_audit_fft2d_f64": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "tfft_audit_forward_rgb8_f64": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tfft_bins_sort": (_i, [_vp, _vp, _u64]),
    "tfft_set_bit_index": (_i, [_vp, _vp, _u64]),
    "tfft_bins_register_dev": (_i, [_vp, _vp, _u64]),
    "tfft_profile_stage": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _u64, _d, C.POINTER(C.c_float), _pi]),
    "tfft_timer_begin": (_i, [_vp]),
    "tfft_timer_end": (_i, [_vp, C.POINTER(C.c_float)]),
}


class TfftError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__("%s failed: %s (%d)" % (where, STATUS.get(status, "?"), status))


def _check(rc, where):
    if rc != TFFT_OK:
        raise TfftError(rc, where)


_libs = {}


def load(path=None):
    """Open the library and bind every declared symbol.  Raises if it is missing:
    there is no fallback implementation."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _libs[path] = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def make_bins(triples):
    """(n,3) int array of (plane,y,x) -> tfft_bin records."""
    t = np.asarray(triples)
    b = np.zeros(len(t), BIN_DTYPE)
    b["plane"], b["y"], b["x"] = t[:, 0], t[:, 1], t[:, 2]
    return b


def bins_to_triples(bins):
    return np.stack([bins["plane"].astype(np.int32), bins["y"].astype(np.int32), bins["x"].astype(np.int32)], axis=1)


class Walk:
    """Host keyed walk (KS + Turtle + density gate), resumable."""

    def __init__(self, key_walk: bytes, ph: int, pw: int, rmin=0.05, rmax=0.45, density=0.7, lib=None):
        self.lib = lib or load()
        self.h = C.c_void_p()
        _check(self.lib.tfft_walk_create(key_walk, ph, pw, rmin, rmax, density, C.byref(self.h)), "tfft_walk_create")
        self.skipped = 0

    def start(self):
        p, y, x = C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.tfft_walk_start(self.h, C.byref(p), C.byref(y), C.byref(x)), "tfft_walk_start")
        return p.value, y.value, x.value

    def next(self, n):
        out = np.zeros(n, BIN_DTYPE)
        sk = C.c_uint64(0)
        _check(self.lib.tfft_walk_next(self.h, n, _ptr(out), C.byref(sk)), "tfft_walk_next")
        self.skipped += sk.value
        return out

    def ks_blocks(self):
        return self.lib.tfft_walk_ks_blocks(self.h)

    def close(self):
        if self.h:
            self.lib.tfft_walk_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def walk_jitter(keys_rgb: bytes, bins, max_jitter, lib=None):
    lib = lib or load()
    out = np.zeros(len(bins), np.float32)
    _check(lib.tfft_walk_jitter(keys_rgb, _ptr(bins), len(bins), max_jitter, _ptr(out)), "tfft_walk_jitter")
    return out


def bins_sort(bins, lib=None):
    """Address-ordered copy of a bin list and bit_index (bit_index[i] = walk position of sorted bin i)."""
    lib = lib or load()
    out = np.ascontiguousarray(bins, BIN_DTYPE).copy()
    idx = np.zeros(len(out), np.uint32)
    _check(lib.tfft_bins_sort(_ptr(out), _ptr(idx), len(out)), "tfft_bins_sort")
    return out, idx


class Context:
    """One tfft_ctx: `slots` resident images of up to max_w x max_h on HIP device `device`."""

    def __init__(self, max_w, max_h, slots=1, device=0, lib=None):
        self.lib = lib or load()
        self.h = C.c_void_p()
        _check(self.lib.tfft_create(device, max_w, max_h, slots, C.byref(self.h)), "tfft_create")
        self.slots = slots

    def close(self):
        if self.h:
            self.lib.tfft_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle: int):
        _check(self.lib.tfft_set_stream(self.h, C.c_void_p(stream_handle)), "tfft_set_stream")

    def sync(self):
        _check(self.lib.tfft_sync(self.h), "tfft_sync")

    def set_bit_index(self, bit_index=None):
        """bins[i] of later embed/read/batch calls carries stream bit bit_index[i]; None clears."""
        if bit_index is None:
            _check(self.lib.tfft_set_bit_index(self.h, None, 0), "tfft_set_bit_index")
            return
        idx = np.ascontiguousarray(bit_index, np.uint32)
        _check(self.lib.tfft_set_bit_index(self.h, _ptr(idx), len(idx)), "tfft_set_bit_index")

    def bins_register_dev(self, bins_ptr, n):
        _check(self.lib.tfft_bins_register_dev(self.h, _ptr(bins_ptr), n), "tfft_bins_register_dev")

    def plan_info(self, w, h, n_images=1):
        info = (C.c_int * 4)()
        _check(self.lib.tfft_plan_info(self.h, w, h, n_images, info), "tfft_plan_info")
        return {"direct": bool(info[0]), "log_n1": info[1], "log_n2": info[2], "fused": bool(info[3]), "two_step": not info[0]}

    def device_bytes(self):
        return self.lib.tfft_device_bytes(self.h)

    # ---- single-image API (host arrays, or device pointers as ints with *_dev) ----------
    def forward_rgb8(self, rgb, center=False, slot=0):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w = rgb.shape[:2]
        pw, ph = C.c_int(), C.c_int()
        _check(self.lib.tfft_forward_rgb8(self.h, slot, _ptr(rgb), w, h, int(center), C.byref(pw), C.byref(ph)),
               "tfft_forward_rgb8")
        return pw.value, ph.value

    def forward_rgb8_dev(self, ptr, w, h, center=False, slot=0):
        pw, ph = C.c_int(), C.c_int()
        _check(self.lib.tfft_forward_rgb8_dev(self.h, slot, _ptr(ptr), w, h, int(center), C.byref(pw), C.byref(ph)),
               "tfft_forward_rgb8_dev")
        return pw.value, ph.value

    def medians(self, slot=0):
        med = np.zeros(3, np.float64)
        _check(self.lib.tfft_medians(self.h, slot, _ptr(med)), "tfft_medians")
        return med

    def exact_info(self):
        """bins per plane the last medians() / capacity() re-evaluated in fp64 (0: the fp32 answer was returned)"""
        out = np.zeros(3, np.int32)
        _check(self.lib.tfft_exact_info(self.h, _ptr(out)), "tfft_exact_info")
        return [int(v) for v in out]

    def median_path(self, slot=0):
        """Per plane: 1 if the last median came from the sampled fast path, 0 if from the full fallback."""
        out = np.zeros(3, np.int32)
        _check(self.lib.tfft_median_path(self.h, slot, _ptr(out)), "tfft_median_path")
        return out

    def capacity(self, thr, rmin=0.05, rmax=0.45, slot=0):
        thr = np.ascontiguousarray(thr, np.float64)
        u = C.c_uint64(0)
        _check(self.lib.tfft_capacity(self.h, slot, rmin, rmax, _ptr(thr), C.byref(u)), "tfft_capacity")
        return u.value

    def lowfreq_mag(self, region, slot=0):
        out = np.zeros((3, region, region), np.float64)
        _check(self.lib.tfft_lowfreq_mag(self.h, slot, region, _ptr(out)), "tfft_lowfreq_mag")
        return out

    def embed_bins(self, bins, bits, alpha=0.5, jitter=None, adaptive=False, med=None, slot=0):
        bits = np.ascontiguousarray(bits, np.uint8)
        jit = np.ascontiguousarray(jitter, np.float32) if jitter is not None else None
        m = np.ascontiguousarray(med, np.float64) if med is not None else None
        _check(self.lib.tfft_embed_bins(self.h, slot, _ptr(bins), _ptr(bits), _ptr(jit), len(bins), alpha,
                                        int(adaptive), _ptr(m)), "tfft_embed_bins")

    def read_bins(self, bins, alpha=0.5, jitter=None, adaptive=False, med=None, slot=0):
        out = np.zeros(len(bins), np.uint8)
        jit = np.ascontiguousarray(jitter, np.float32) if jitter is not None else None
        m = np.ascontiguousarray(med, np.float64) if med is not None else None
        _check(self.lib.tfft_read_bins(self.h, slot, _ptr(bins), _ptr(jit), len(bins), alpha, int(adaptive),
                                       _ptr(m), _ptr(out)), "tfft_read_bins")
        return out

    def inverse_rgb8(self, w, h, slot=0):
        out = np.zeros((h, w, 3), np.uint8)
        _check(self.lib.tfft_inverse_rgb8(self.h, slot, _ptr(out)), "tfft_inverse_rgb8")
        return out

    def inverse_rgb8_dev(self, ptr, slot=0):
        _check(self.lib.tfft_inverse_rgb8_dev(self.h, slot, _ptr(ptr)), "tfft_inverse_rgb8_dev")

    def embed_bins_dev(self, bins_ptr, bits_ptr, n, alpha=0.5, slot=0):
        _check(self.lib.tfft_embed_bins_dev(self.h, slot, _ptr(bins_ptr), _ptr(bits_ptr), None, n, alpha, 0, None),
               "tfft_embed_bins_dev")

    def read_bins_dev(self, bins_ptr, n, out_ptr, alpha=0.5, slot=0):
        _check(self.lib.tfft_read_bins_dev(self.h, slot, _ptr(bins_ptr), None, n, alpha, 0, None, _ptr(out_ptr)),
               "tfft_read_bins_dev")

    # ---- (f-4) fp64 audit transform: the reference's fft2d in double on the device ----------
    def audit_fft2d_f64(self, planes, inverse=False):
        """planes: (n, PH, PW) complex128; returns the transformed copy."""
        a = np.ascontiguousarray(planes, np.complex128).copy()
        n, ph, pw = a.shape
        _check(self.lib.tfft_audit_fft2d_f64(self.h, _ptr(a), n, ph, pw, int(inverse)), "tfft_audit_fft2d_f64")
        return a

    def audit_forward_rgb8_f64(self, rgb, center=False):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w = rgb.shape[:2]
        ph, pw = 1 << (h - 1).bit_length(), 1 << (w - 1).bit_length()
        out = np.zeros((3, ph, pw), np.complex128)
        _check(self.lib.tfft_audit_forward_rgb8_f64(self.h, _ptr(rgb), w, h, int(center), _ptr(out)), "tfft_audit_forward_rgb8_f64")
        return out

    def download_spectrum(self, pw, ph, slot=0):
        out = np.zeros((3, ph, pw), np.complex64)
        _check(self.lib.tfft_download_spectrum(self.h, slot, _ptr(out)), "tfft_download_spectrum")
        return out

    # ---- batches (device pointers) ---------------------------------------------------------
    def embed_batch_dev(self, n_images, rgb_ptr, w, h, bins_ptr, bits_ptr, n_bits, out_ptr, alpha=0.5, center=False,
                        rmin=0.05, rmax=0.45, magmin=0.01, usable_ptr=None):
        _check(self.lib.tfft_embed_batch_dev(self.h, n_images, _ptr(rgb_ptr), w, h, int(center), _ptr(bins_ptr),
                                             _ptr(bits_ptr), n_bits, alpha, rmin, rmax, magmin, _ptr(usable_ptr),
                                             _ptr(out_ptr)), "tfft_embed_batch_dev")

    def extract_batch_dev(self, n_images, rgb_ptr, w, h, bins_ptr, n_bits, bits_out_ptr, alpha=0.5, center=False):
        _check(self.lib.tfft_extract_batch_dev(self.h, n_images, _ptr(rgb_ptr), w, h, int(center), _ptr(bins_ptr),
                                               n_bits, alpha, _ptr(bits_out_ptr)), "tfft_extract_batch_dev")

    def embed_stream_batch_dev(self, n_images, rgb_ptr, w, h, bins_ptr, n_bins, header_ptr, payload_ptr, payload_len, out_ptr,
                               alpha=0.5, center=False, rmin=0.05, rmax=0.45, magmin=0.01, usable_ptr=None):
        _check(self.lib.tfft_embed_stream_batch_dev(self.h, n_images, _ptr(rgb_ptr), w, h, int(center), _ptr(bins_ptr), n_bins,
                                                    _ptr(header_ptr), _ptr(payload_ptr), payload_len, alpha, rmin, rmax, magmin,
                                                    _ptr(usable_ptr), _ptr(out_ptr)), "tfft_embed_stream_batch_dev")

    def extract_stream_batch_dev(self, n_images, rgb_ptr, w, h, bins_ptr, n_bins, header_out_ptr, payload_out_ptr, max_payload_len,
                                 status_out_ptr, raw_bits_out_ptr=None, alpha=0.5, center=False):
        _check(self.lib.tfft_extract_stream_batch_dev(self.h, n_images, _ptr(rgb_ptr), w, h, int(center), _ptr(bins_ptr), n_bins, alpha,
                                                      _ptr(header_out_ptr), _ptr(payload_out_ptr), max_payload_len, _ptr(status_out_ptr),
                                                      _ptr(raw_bits_out_ptr)), "tfft_extract_stream_batch_dev")

    STAGES = ["rows_fwd", "cols_fwd_a", "cols_fwd_b", "embed", "cols_inv_a", "cols_inv_b", "rows_inv", "read",
              "medians", "capacity", "cols_fwd_read"]

    def profile_stage(self, stage, reps, rgb_ptr=None, out_ptr=None, bins_ptr=None, bits_ptr=None, bits_out_ptr=None,
                      n_bits=0, alpha=0.5, n_images=1):
        """Mean ms of one repetition of pipeline stage `stage` over slots [0, n_images) (HIP events on
        the context stream) and the number of kernel launches per repetition."""
        ms, nl = C.c_float(0), C.c_int(0)
        _check(self.lib.tfft_profile_stage(self.h, n_images, stage, reps, _ptr(rgb_ptr), _ptr(out_ptr), _ptr(bins_ptr),
                                           _ptr(bits_ptr), _ptr(bits_out_ptr), n_bits, alpha, C.byref(ms),
                                           C.byref(nl)), "tfft_profile_stage")
        return ms.value, nl.value

    def embed_batch_host(self, rgb, bins, bits, out, usable=None, alpha=0.5, center=False, rmin=0.05, rmax=0.45,
                         magmin=0.01):
        """rgb/out: (n,H,W,3) uint8 host arrays (pinned for overlap); bits: (n,n_bits) uint8."""
        n, h, w = rgb.shape[:3]
        _check(self.lib.tfft_embed_batch(self.h, n, _ptr(rgb), w, h, int(center), _ptr(bins), _ptr(bits), bits.shape[1], alpha,
                                         rmin, rmax, magmin, _ptr(usable), _ptr(out)), "tfft_embed_batch")

    def extract_batch_host(self, rgb, bins, bits_out, alpha=0.5, center=False):
        n, h, w = rgb.shape[:3]
        _check(self.lib.tfft_extract_batch(self.h, n, _ptr(rgb), w, h, int(center), _ptr(bins), bits_out.shape[1], alpha,
                                           _ptr(bits_out)), "tfft_extract_batch")

    def embed_stream_batch_host(self, rgb, bins, header, payload, out, usable=None, alpha=0.5, center=False, rmin=0.05, rmax=0.45, magmin=0.01):
        n, h, w = rgb.shape[:3]
        _check(self.lib.tfft_embed_stream_batch(self.h, n, _ptr(rgb), w, h, int(center), _ptr(bins), len(bins), _ptr(header), _ptr(payload),
                                                payload.shape[1] if payload is not None else 0, alpha, rmin, rmax, magmin, _ptr(usable), _ptr(out)),
               "tfft_embed_stream_batch")

    def extract_stream_batch_host(self, rgb, bins, header_out, payload_out, status_out, raw_bits_out=None, alpha=0.5, center=False):
        n, h, w = rgb.shape[:3]
        _check(self.lib.tfft_extract_stream_batch(self.h, n, _ptr(rgb), w, h, int(center), _ptr(bins), len(bins), alpha, _ptr(header_out),
                                                  _ptr(payload_out), payload_out.shape[1], _ptr(status_out), _ptr(raw_bits_out)),
               "tfft_extract_stream_batch")

    def frame_expand_dev(self, n_images, header_ptr, payload_ptr, payload_len, bits_out_ptr):
        _check(self.lib.tfft_frame_expand_dev(self.h, n_images, _ptr(header_ptr), _ptr(payload_ptr), payload_len,
                                              _ptr(bits_out_ptr)), "tfft_frame_expand_dev")

    def frame_majority_dev(self, n_images, bits_ptr, payload_len, header_out_ptr, payload_out_ptr):
        _check(self.lib.tfft_frame_majority_dev(self.h, n_images, _ptr(bits_ptr), payload_len, _ptr(header_out_ptr),
                                                _ptr(payload_out_ptr)), "tfft_frame_majority_dev")

    def timer_begin(self):
        _check(self.lib.tfft_timer_begin(self.h), "tfft_timer_begin")

    def timer_end(self):
        ms = C.c_float(0)
        _check(self.lib.tfft_timer_end(self.h, C.byref(ms)), "tfft_timer_end")
        return ms.value


# ---- PNG files -> GPU -> PNG files (libtfpipe.so, include/turtlefft_pipe.h) ---------------------------------------------
_PIPE = None


def load_pipe():
    """libtfpipe.so next to the device library (raises when it is missing, like load())"""
    global _PIPE
    if _PIPE is None:
        path = os.path.join(_PKG, "libtfpipe.so")
        if not os.path.exists(path):
            raise RuntimeError("libtfpipe.so is not built (make -C steganosaurus_amd/csrc)")
        lib = C.CDLL(path)
        lib.tfp_embed_png_batch.restype = C.c_int
        lib.tfp_embed_png_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]
        lib.tfp_extract_png_batch.restype = C.c_int
        lib.tfp_extract_png_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _PIPE = lib
    return _PIPE


def _paths(paths):
    arr = (C.c_char_p * len(paths))()
    arr[:] = [p.encode() for p in paths]
    return arr


def embed_png_batch(ctx, in_paths, out_paths, w, h, bins, headers, payloads, chunk=16, threads=8, png_level=1, center=False, alpha=0.5,
                    rmin=0.05, rmax=0.45, magmin=0.01, want_usable=True):
    """tfp_embed_png_batch; returns (usable per file or None, [wall, decode, device, encode] ms)"""
    lib = load_pipe()
    n = len(in_paths)
    headers = np.ascontiguousarray(headers, np.uint8); payloads = np.ascontiguousarray(payloads, np.uint8)
    assert headers.shape == (n, 38) and payloads.shape[0] == n
    usable = np.zeros(n, np.uint64) if want_usable else None
    ms = np.zeros(4, np.float64)
    rc = lib.tfp_embed_png_batch(ctx.h, n, _paths(in_paths), _paths(out_paths), w, h, int(center), _ptr(bins), len(bins), _ptr(headers), _ptr(payloads),
                                 payloads.shape[1], alpha, rmin, rmax, magmin, chunk, threads, png_level, _ptr(usable) if want_usable else None, _ptr(ms))
    _check(rc, "tfp_embed_png_batch")
    return usable, ms


def extract_png_batch(ctx, in_paths, w, h, bins, max_payload_len, chunk=16, threads=8, center=False, alpha=0.5):
    """tfp_extract_png_batch; returns (headers (n, 38), payloads (n, max_payload_len), status (n,), ms)"""
    lib = load_pipe()
    n = len(in_paths)
    hdr = np.zeros((n, 38), np.uint8); pay = np.zeros((n, max_payload_len), np.uint8); st = np.zeros(n, np.int32); ms = np.zeros(4, np.float64)
    rc = lib.tfp_extract_png_batch(ctx.h, n, _paths(in_paths), w, h, int(center), _ptr(bins), len(bins), alpha, _ptr(hdr), _ptr(pay), max_payload_len,
                                   _ptr(st), chunk, threads, _ptr(ms))
    _check(rc, "tfp_extract_png_batch")
    return hdr, pay, st, ms

