// tfft_walk.cpp -- host side of the hot path: keyed bin-index materialisation.
//
// Replaces KS (steganosaur.cpp:665-695), Turtle (S:749-810) and the density
// loop around advance_to_valid (S:1076-1081, S:1206).  The walk is a pure
// function of (key_walk, PH, PW, rmin, rmax, density) -- the magnitude test is
// commented out in the reference (S:797-799) -- and is strictly sequential
// (every accepted position depends on every previous opcode), so it stays on
// the host, is computed once and is shared by all images of a batch.
//
// Differences in mechanism, not in results: flat bit-packed visited map
// instead of vector<vector<vector<uint8_t>>>, radius test on exact integers
// instead of hypot() per step, one SHA-256 compression per keystream block
// with a pre-padded message, and a bounded search that returns
// TFFT_E_EXHAUSTED where the reference spins forever.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/turtlefft_hip.h"

namespace {

// ------------------------------------------------------------------ SHA-256 (FIPS 180-4)
const uint32_t K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
const uint32_t IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

void compress(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    memcpy(h, IV, sizeof h);
    size_t off = 0;
    for (; off + 64 <= len; off += 64) compress(h, data + off);
    uint8_t tail[128] = {0};
    const size_t rem = len - off;
    memcpy(tail, data + off, rem);
    tail[rem] = 0x80;
    const size_t tl = (rem + 9 <= 64) ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    compress(h, tail);
    if (tl == 128) compress(h, tail + 64);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}

// ------------------------------------------------------------------ KS (S:665-695)
// block k = SHA256(key[32] || 0xAA || le32(k)): a 37-byte message = one compression
struct KeyStream {
    uint8_t msg[64];      // pre-padded single block
    uint8_t state[32];
    int pos = 32;
    uint32_t ctr = 0;
    uint32_t pool = 0;
    int nbits = 0;
    explicit KeyStream(const uint8_t key[32]) {
        memset(msg, 0, sizeof msg);
        memcpy(msg, key, 32);
        msg[32] = 0xAA;
        msg[37] = 0x80;
        msg[62] = (37 * 8) >> 8; msg[63] = (37 * 8) & 0xFF;
    }
    inline uint8_t next_byte() {                       // S:673-684
        if (pos >= 32) {
            msg[33] = (uint8_t)ctr; msg[34] = (uint8_t)(ctr >> 8); msg[35] = (uint8_t)(ctr >> 16); msg[36] = (uint8_t)(ctr >> 24);
            uint32_t h[8];
            memcpy(h, IV, sizeof h);
            compress(h, msg);
            for (int i = 0; i < 8; i++) { state[4 * i] = h[i] >> 24; state[4 * i + 1] = h[i] >> 16; state[4 * i + 2] = h[i] >> 8; state[4 * i + 3] = h[i]; }
            pos = 0; ctr++;
        }
        return state[pos++];
    }
    inline int next_opcode3() {                        // S:685 (MSB first, leftover bits persist)
        while (nbits < 3) { pool = (pool << 8) | next_byte(); nbits += 8; }
        const int op = (pool >> (nbits - 3)) & 7;
        nbits -= 3;
        return op;
    }
    inline float jitter(double maxj) {                 // S:690-694 (high byte first)
        const int hi = next_byte(), lo = next_byte();
        const int16_t r = (int16_t)((hi << 8) | lo);
        return (float)((r / 32768.0) * maxj);
    }
};

// s_lo <= y*y + x*x <= s_hi  <=>  lo <= hypot(y,x) <= hi in the reference's double arithmetic
// (sqrt of an exactly representable integer is correctly rounded, like glibc's hypot)
void radius_bounds(double lo, double hi, uint64_t* s_lo, uint64_t* s_hi, bool* empty) {
    *empty = !(hi >= 0.0) || !(lo <= hi);
    if (*empty) { *s_lo = 1; *s_hi = 0; return; }
    uint64_t c = 0;
    if (lo > 0.0) {
        c = (uint64_t)floor(lo * lo);
        while (c > 0 && sqrt((double)c) >= lo) c--;
        while (sqrt((double)c) < lo) c++;
    }
    *s_lo = c;
    c = (uint64_t)floor(hi * hi) + 2;
    while (c > 0 && sqrt((double)c) > hi) c--;
    *s_hi = c;
    if (sqrt((double)c) > hi) *empty = true;
}

}  // namespace

struct tfft_walk {
    int H, W;
    int y, x, plane;
    int start[3];
    uint8_t dens_thr;
    uint64_t s_lo, s_hi;
    bool ring_empty;
    KeyStream ks;
    std::vector<uint64_t> visited;      // 3*H*W bits
    explicit tfft_walk(const uint8_t key[32]) : ks(key) {}

    inline size_t bit(int p, int yy, int xx) const { return ((size_t)p * H + yy) * W + xx; }
    inline bool seen(size_t b) const { return (visited[b >> 6] >> (b & 63)) & 1; }
    inline void mark(size_t b) { visited[b >> 6] |= (1ull << (b & 63)); }
    inline bool on_axis(int yy, int xx) const {                       // S:698-700
        return yy == 0 || xx == 0 || (H % 2 == 0 && yy == H / 2) || (W % 2 == 0 && xx == W / 2);
    }
    inline bool in_ring(int yy, int xx) const {                       // S:771-774
        const uint64_t s = (uint64_t)yy * yy + (uint64_t)xx * xx;
        return !ring_empty && s >= s_lo && s <= s_hi;
    }
    inline void mirror(int yy, int xx, int* cy, int* cx) const {      // S:370-372
        *cy = (yy == 0) ? 0 : H - yy;
        *cx = (xx == 0) ? 0 : W - xx;
    }
    bool acceptable(int p, int yy, int xx) const {
        if (on_axis(yy, xx) || !in_ring(yy, xx) || seen(bit(p, yy, xx))) return false;
        int cy, cx; mirror(yy, xx, &cy, &cx);
        return !seen(bit(p, cy, cx));
    }
    bool any_left() const {
        for (int p = 0; p < 3; p++)
            for (int yy = 0; yy < H; yy++)
                for (int xx = 0; xx < W; xx++)
                    if (acceptable(p, yy, xx)) return true;
        return false;
    }
    // Turtle::advance_to_valid S:778-804
    int advance() {
        const uint64_t patience = 16ull * 3 * (uint64_t)H * W + (1u << 20);
        uint64_t streak = 0;
        for (;;) {
            switch (ks.next_opcode3()) {
                case 0: plane = (plane + 1) % 3; break;
                case 1: x = (x + 1 == W) ? 0 : x + 1; break;
                case 2: y = (y + 1 == H) ? 0 : y + 1; break;
                case 3: x = (x == 0) ? W - 1 : x - 1; break;
                case 4: y = (y == 0) ? H - 1 : y - 1; break;
                case 5: x = (x + 1 == W) ? 0 : x + 1; y = (y + 1 == H) ? 0 : y + 1; break;
                case 6: x = (x == 0) ? W - 1 : x - 1; y = (y + 1 == H) ? 0 : y + 1; break;
                default: break;
            }
            if (acceptable(plane, y, x)) return TFFT_OK;
            if (++streak >= patience) {
                if (!any_left()) return TFFT_E_EXHAUSTED;
                streak = 0;
            }
        }
    }
    void mark_here() {                                                // S:805-809
        mark(bit(plane, y, x));
        int cy, cx; mirror(y, x, &cy, &cx);
        mark(bit(plane, cy, cx));
    }
};

extern "C" {

int tfft_walk_create(const uint8_t key_walk[32], int ph, int pw, double rmin, double rmax, double density,
                     tfft_walk** out) {
    if (!key_walk || !out || ph < 1 || pw < 1 || ph > 65536 || pw > 65536) return TFFT_E_INVALID;
    tfft_walk* w = new (std::nothrow) tfft_walk(key_walk);
    if (!w) return TFFT_E_NOMEM;
    w->H = ph; w->W = pw;
    try {
        w->visited.assign(((size_t)3 * ph * pw + 63) / 64, 0);
    } catch (...) { delete w; return TFFT_E_NOMEM; }
    const int mn = ph < pw ? ph : pw;
    radius_bounds(rmin * mn, rmax * mn, &w->s_lo, &w->s_hi, &w->ring_empty);
    // (uint8_t)floor(density*256.0) S:688: out-of-range values wrap modulo 256 on x86-64
    w->dens_thr = (uint8_t)(long long)floor(density * 256.0);
    // Turtle ctor S:762-770
    char pre[64];
    const int n = snprintf(pre, sizeof pre, "seed:%dx%d|key:", ph, pw);
    uint8_t buf[96], h[32];
    memcpy(buf, pre, (size_t)n);
    memcpy(buf + n, key_walk, 32);
    sha256(buf, (size_t)n + 32, h);
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s = (s << 8) | h[i];
    w->y = (int)(s % (uint64_t)ph);
    w->x = (int)((s >> 16) % (uint64_t)pw);
    w->plane = (int)((s >> 32) % 3);
    w->start[0] = w->plane; w->start[1] = w->y; w->start[2] = w->x;
    *out = w;
    return TFFT_OK;
}

int tfft_walk_next(tfft_walk* w, uint64_t n, tfft_bin* out, uint64_t* skipped) {
    if (!w || (n && !out)) return TFFT_E_INVALID;
    for (uint64_t i = 0; i < n; i++) {
        for (;;) {                                                    // S:1076-1081
            const int rc = w->advance();
            if (rc != TFFT_OK) return rc;
            if (w->ks.next_byte() < w->dens_thr) break;               // KS::hit_density S:686-689
            w->mark_here();
            if (skipped) (*skipped)++;
        }
        out[i].x = (uint16_t)w->x; out[i].y = (uint16_t)w->y; out[i].plane = (uint8_t)w->plane;
        out[i].rsv[0] = out[i].rsv[1] = out[i].rsv[2] = 0;
        w->mark_here();
    }
    return TFFT_OK;
}

int tfft_walk_start(const tfft_walk* w, int* plane, int* y, int* x) {
    if (!w) return TFFT_E_INVALID;
    if (plane) *plane = w->start[0];
    if (y) *y = w->start[1];
    if (x) *x = w->start[2];
    return TFFT_OK;
}

uint32_t tfft_walk_ks_blocks(const tfft_walk* w) { return w ? w->ks.ctr : 0; }

int tfft_walk_destroy(tfft_walk* w) {
    delete w;
    return TFFT_OK;
}

int tfft_walk_jitter(const uint8_t keys_rgb[96], const tfft_bin* bins, uint64_t n, double max_jitter, float* out) {
    if (!keys_rgb || (n && (!bins || !out))) return TFFT_E_INVALID;
    KeyStream ks[3] = {KeyStream(keys_rgb), KeyStream(keys_rgb + 32), KeyStream(keys_rgb + 64)};
    for (uint64_t i = 0; i < n; i++) {
        if (bins[i].plane > 2) return TFFT_E_BIN_RANGE;
        out[i] = ks[bins[i].plane].jitter(max_jitter);                // S:719, S:1208
    }
    return TFFT_OK;
}

// Address order for the device kernels: stable LSD counting sort on (plane, y, x).  The walk is a
// pseudo-random tour of the annulus, so in walk order every bin costs its own DRAM row activation;
// visiting the same set in row-major order lets neighbouring lanes share rows and sectors.
int tfft_bins_sort(tfft_bin* bins, uint32_t* bit_index, uint64_t n) {
    if (n && (!bins || !bit_index)) return TFFT_E_INVALID;
    if (n > 0xFFFFFFFFull) return TFFT_E_TOO_LARGE;
    std::vector<tfft_bin> tb(n);
    std::vector<uint32_t> ti(n);
    for (uint64_t i = 0; i < n; i++) bit_index[i] = (uint32_t)i;
    std::vector<uint64_t> cnt(65537);
    auto pass = [&](auto key, const tfft_bin* sb, const uint32_t* si, tfft_bin* db, uint32_t* di) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (uint64_t i = 0; i < n; i++) cnt[(size_t)key(sb[i]) + 1]++;
        for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
        for (uint64_t i = 0; i < n; i++) { const uint64_t d = cnt[key(sb[i])]++; db[d] = sb[i]; di[d] = si[i]; }
    };
    pass([](const tfft_bin& b) { return (unsigned)b.x; }, bins, bit_index, tb.data(), ti.data());
    pass([](const tfft_bin& b) { return (unsigned)b.y; }, tb.data(), ti.data(), bins, bit_index);
    pass([](const tfft_bin& b) { return (unsigned)b.plane; }, bins, bit_index, tb.data(), ti.data());
    std::copy(tb.begin(), tb.end(), bins);
    std::copy(ti.begin(), ti.end(), bit_index);
    return TFFT_OK;
}

// exported for tfft_capi: the same integer radius bounds drive the capacity kernel
void tfft_internal_radius_bounds(double lo, double hi, uint64_t* s_lo, uint64_t* s_hi, int* empty) {
    bool e;
    radius_bounds(lo, hi, s_lo, s_hi, &e);
    *empty = e ? 1 : 0;
}

}  // extern "C"
