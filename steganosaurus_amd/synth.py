"""Synthetic covers and secrets shared by tests, bench.py and the golden generator.

The generator is the one SURVEY.md section 8(d) fixes for every measurement:
each byte is ``128 + ((lcg >> 24) % 64) - 32`` with
``lcg = lcg * 1664525 + 1013904223 (mod 2**32)``, seed ``12345 + image_index``
(textured mid-grey, 96..159, never clamps much).
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_bytes(n: int, seed: int) -> np.ndarray:
    """First ``n`` bytes of the stream (state advanced before each byte)."""
    out = np.empty(n, dtype=np.uint8)
    state = np.uint32(seed & 0xFFFFFFFF)
    chunk = 1 << 22
    with np.errstate(over="ignore"):
        # x_k = a^k x_0 + c (a^(k-1) + ... + 1)  (mod 2^32), k = 1..m
        m = min(chunk, n)
        ak = np.cumprod(np.full(m, _A, dtype=np.uint32), dtype=np.uint32)        # a^1..a^m
        geo = np.cumsum(np.concatenate(([np.uint32(1)], ak[:-1])), dtype=np.uint32)  # 1+a+..+a^(k-1)
        pos = 0
        while pos < n:
            k = min(m, n - pos)
            x = ak[:k] * state + _C * geo[:k]
            out[pos:pos + k] = (128 + ((x >> np.uint32(24)) % np.uint32(64)).astype(np.int32) - 32).astype(np.uint8)
            state = x[k - 1]
            pos += k
    return out


def cover_rgb(width: int, height: int, index: int = 0) -> np.ndarray:
    """H x W x 3 uint8 synthetic cover number ``index``."""
    return lcg_bytes(width * height * 3, 12345 + index).reshape(height, width, 3)


def gradient_cover(width: int, height: int, seed: int = 1) -> np.ndarray:
    """Gradient + noise cover in the spirit of the reference's gen_png tool
    (tools/gen_png.cpp:5-21): smooth content, so clamping/rounding noise matters."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    img = np.stack([(x * 255) // max(1, width - 1), (y * 255) // max(1, height - 1),
                    ((x + y) * 255) // max(1, width + height - 2)], axis=-1).astype(np.int32)
    img = img // 2 + 64 + rng.integers(-8, 9, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def secret_ascii(n: int, seed: int = 7) -> bytes:
    """``n`` bytes of seeded printable ASCII (no NUL: the reference prints with %s)."""
    rng = np.random.default_rng(seed)
    return bytes(rng.integers(0x20, 0x7F, size=n, dtype=np.uint8).tolist())


def n_stream_bits(secret_len: int) -> int:
    """Rep-3(38-byte header) + Rep-7(ciphertext + 16-byte tag): S:903, S:986-995."""
    return 38 * 8 * 3 + (secret_len + 16) * 8 * 7
