// tf_crypto.cpp -- see tf_crypto.h
#include "tf_crypto.h"

#include <stdio.h>
#include <string.h>
#include <sys/random.h>

namespace tfh {

void secure_zero(void* p, size_t n) {
    volatile uint8_t* v = static_cast<volatile uint8_t*>(p);
    while (n--) *v++ = 0;
}

// ------------------------------------------------------------------ SHA-256
namespace {
const uint32_t K[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
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
}  // namespace

Sha256::Sha256() : len(0), fill(0) {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                                   0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, iv, sizeof iv);
}
void Sha256::update(const void* data, size_t n) {
    const uint8_t* d = static_cast<const uint8_t*>(data);
    len += n;
    while (n) {
        size_t k = 64 - fill;
        if (k > n) k = n;
        memcpy(buf + fill, d, k);
        fill += k; d += k; n -= k;
        if (fill == 64) { compress(h, buf); fill = 0; }
    }
}
void Sha256::final(uint8_t out[32]) {
    const uint64_t bits = len * 8;
    const uint8_t pad = 0x80, zero = 0;
    update(&pad, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (8 * (7 - i)));
    update(lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}
std::array<uint8_t, 32> sha256(const void* data, size_t n) {
    Sha256 s; s.update(data, n);
    std::array<uint8_t, 32> out; s.final(out.data());
    return out;
}

// ------------------------------------------------------------------ HMAC / PBKDF2 / HKDF
namespace {
struct HmacKey {     // inner/outer states after absorbing the padded key: 2 compressions per HMAC of a short message
    Sha256 inner, outer;
    HmacKey(const uint8_t* key, size_t klen) {
        uint8_t k0[64] = {0}, pad[64];
        if (klen > 64) { auto d = sha256(key, klen); memcpy(k0, d.data(), 32); } else memcpy(k0, key, klen);
        for (int i = 0; i < 64; i++) pad[i] = k0[i] ^ 0x36;
        inner.update(pad, 64);
        for (int i = 0; i < 64; i++) pad[i] = k0[i] ^ 0x5c;
        outer.update(pad, 64);
        secure_zero(k0, sizeof k0); secure_zero(pad, sizeof pad);
    }
    void mac(const uint8_t* msg, size_t mlen, uint8_t out[32]) const {
        Sha256 i = inner; i.update(msg, mlen);
        uint8_t ih[32]; i.final(ih);
        Sha256 o = outer; o.update(ih, 32); o.final(out);
    }
};
}  // namespace

void hmac_sha256(const uint8_t* key, size_t klen, const uint8_t* msg, size_t mlen, uint8_t out[32]) {
    HmacKey(key, klen).mac(msg, mlen, out);
}

void pbkdf2_hmac_sha256(const uint8_t* pass, size_t plen, const uint8_t* salt, size_t slen, uint32_t iters,
                        uint8_t* out, size_t dklen) {
    const HmacKey hk(pass, plen);
    const uint32_t blocks = (uint32_t)((dklen + 31) / 32);
    std::vector<uint8_t> msg(slen + 4);
    if (slen) memcpy(msg.data(), salt, slen);
    for (uint32_t i = 1; i <= blocks; i++) {
        msg[slen] = (uint8_t)(i >> 24); msg[slen + 1] = (uint8_t)(i >> 16); msg[slen + 2] = (uint8_t)(i >> 8); msg[slen + 3] = (uint8_t)i;
        uint8_t U[32], T[32];
        hk.mac(msg.data(), msg.size(), U);
        memcpy(T, U, 32);
        for (uint32_t j = 2; j <= iters; j++) {
            hk.mac(U, 32, U);
            for (int k = 0; k < 32; k++) T[k] ^= U[k];
        }
        const size_t off = (size_t)(i - 1) * 32, need = dklen - off < 32 ? dklen - off : 32;
        memcpy(out + off, T, need);
        secure_zero(U, 32); secure_zero(T, 32);
    }
}

void hkdf_extract(const uint8_t* salt, size_t slen, const uint8_t* ikm, size_t ilen, uint8_t prk[32]) {
    hmac_sha256(salt, slen, ikm, ilen, prk);     // salt == NULL/0: HMAC key of 64 zero bytes
}
void hkdf_expand(const uint8_t prk[32], const uint8_t* info, size_t ilen, uint8_t* out, size_t L) {
    uint8_t T[32]; size_t tlen = 0, pos = 0; uint8_t ctr = 1;
    std::vector<uint8_t> msg;
    while (pos < L) {
        msg.assign(T, T + tlen);
        msg.insert(msg.end(), info, info + ilen);
        msg.push_back(ctr);
        hmac_sha256(prk, 32, msg.data(), msg.size(), T);
        tlen = 32;
        const size_t need = L - pos < 32 ? L - pos : 32;
        memcpy(out + pos, T, need);
        pos += need; ctr++;
    }
    secure_zero(T, 32);
}

// ------------------------------------------------------------------ ChaCha20 (RFC 8439 2.3)
namespace {
inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
inline uint32_t ld32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void st32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
#define TFH_QR(a, b, c, d) a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12); a += b; d ^= a; d = rotl(d, 8); c += d; b ^= c; b = rotl(b, 7);
void chacha_block(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, uint8_t out[64]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = ld32(key + 4 * i);
    s[12] = counter; s[13] = ld32(nonce); s[14] = ld32(nonce + 4); s[15] = ld32(nonce + 8);
    uint32_t x[16]; memcpy(x, s, sizeof x);
    for (int i = 0; i < 10; i++) {
        TFH_QR(x[0], x[4], x[8], x[12]) TFH_QR(x[1], x[5], x[9], x[13]) TFH_QR(x[2], x[6], x[10], x[14]) TFH_QR(x[3], x[7], x[11], x[15])
        TFH_QR(x[0], x[5], x[10], x[15]) TFH_QR(x[1], x[6], x[11], x[12]) TFH_QR(x[2], x[7], x[8], x[13]) TFH_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) st32(out + 4 * i, x[i] + s[i]);
}
void chacha_xor(const uint8_t key[32], const uint8_t nonce[12], uint32_t counter, const uint8_t* in, uint8_t* out, size_t len) {
    uint8_t ks[64];
    for (size_t off = 0; off < len; off += 64) {
        chacha_block(key, nonce, counter++, ks);
        const size_t n = len - off < 64 ? len - off : 64;
        for (size_t i = 0; i < n; i++) out[off + i] = in[off + i] ^ ks[i];
    }
    secure_zero(ks, sizeof ks);
}

// ------------------------------------------------------------------ Poly1305 (RFC 8439 2.5), 64-bit limbs via __int128
struct Poly1305 {
    uint64_t r0, r1, r2, s1, s2, h0, h1, h2;   // r and h in 44/44/42-bit limbs
    uint64_t pad0, pad1;
    uint8_t buf[16]; size_t fill;
    explicit Poly1305(const uint8_t key[32]) : h0(0), h1(0), h2(0), fill(0) {
        uint64_t t0 = (uint64_t)ld32(key) | ((uint64_t)ld32(key + 4) << 32);
        uint64_t t1 = (uint64_t)ld32(key + 8) | ((uint64_t)ld32(key + 12) << 32);
        r0 = t0 & 0xffc0fffffffULL;
        r1 = ((t0 >> 44) | (t1 << 20)) & 0xfffffc0ffffULL;
        r2 = (t1 >> 24) & 0x00ffffffc0fULL;
        s1 = r1 * 20; s2 = r2 * 20;
        pad0 = (uint64_t)ld32(key + 16) | ((uint64_t)ld32(key + 20) << 32);
        pad1 = (uint64_t)ld32(key + 24) | ((uint64_t)ld32(key + 28) << 32);
    }
    void block(const uint8_t* m, uint64_t hibit) {
        typedef unsigned __int128 u128;
        const uint64_t t0 = (uint64_t)ld32(m) | ((uint64_t)ld32(m + 4) << 32);
        const uint64_t t1 = (uint64_t)ld32(m + 8) | ((uint64_t)ld32(m + 12) << 32);
        h0 += t0 & 0xfffffffffffULL;
        h1 += ((t0 >> 44) | (t1 << 20)) & 0xfffffffffffULL;
        h2 += ((t1 >> 24) & 0x3ffffffffffULL) | hibit;
        u128 d0 = (u128)h0 * r0 + (u128)h1 * s2 + (u128)h2 * s1;
        u128 d1 = (u128)h0 * r1 + (u128)h1 * r0 + (u128)h2 * s2;
        u128 d2 = (u128)h0 * r2 + (u128)h1 * r1 + (u128)h2 * r0;
        uint64_t c = (uint64_t)(d0 >> 44); h0 = (uint64_t)d0 & 0xfffffffffffULL;
        d1 += c; c = (uint64_t)(d1 >> 44); h1 = (uint64_t)d1 & 0xfffffffffffULL;
        d2 += c; c = (uint64_t)(d2 >> 42); h2 = (uint64_t)d2 & 0x3ffffffffffULL;
        h0 += c * 5; c = h0 >> 44; h0 &= 0xfffffffffffULL;
        h1 += c;
    }
    void update(const uint8_t* m, size_t n) {
        while (n) {
            size_t k = 16 - fill; if (k > n) k = n;
            memcpy(buf + fill, m, k); fill += k; m += k; n -= k;
            if (fill == 16) { block(buf, 1ULL << 40); fill = 0; }
        }
    }
    void pad16() { static const uint8_t z[16] = {0}; if (fill) update(z, 16 - fill); }
    void final(uint8_t tag[16]) {
        if (fill) {      // not used by the AEAD (always padded), kept for a correct standalone MAC
            buf[fill] = 1; for (size_t i = fill + 1; i < 16; i++) buf[i] = 0;
            block(buf, 0);
        }
        uint64_t c = h1 >> 44; h1 &= 0xfffffffffffULL;
        h2 += c; c = h2 >> 42; h2 &= 0x3ffffffffffULL;
        h0 += c * 5; c = h0 >> 44; h0 &= 0xfffffffffffULL;
        h1 += c; c = h1 >> 44; h1 &= 0xfffffffffffULL;
        h2 += c; c = h2 >> 42; h2 &= 0x3ffffffffffULL;
        h0 += c * 5; c = h0 >> 44; h0 &= 0xfffffffffffULL;
        h1 += c;
        uint64_t g0 = h0 + 5; c = g0 >> 44; g0 &= 0xfffffffffffULL;
        uint64_t g1 = h1 + c; c = g1 >> 44; g1 &= 0xfffffffffffULL;
        uint64_t g2 = h2 + c - (1ULL << 42);
        const uint64_t mask = (g2 >> 63) - 1;           // all ones if h >= p
        h0 = (h0 & ~mask) | (g0 & mask); h1 = (h1 & ~mask) | (g1 & mask); h2 = (h2 & ~mask) | (g2 & mask);
        const uint64_t t0 = pad0, t1 = pad1;
        h0 += t0 & 0xfffffffffffULL; c = h0 >> 44; h0 &= 0xfffffffffffULL;
        h1 += (((t0 >> 44) | (t1 << 20)) & 0xfffffffffffULL) + c; c = h1 >> 44; h1 &= 0xfffffffffffULL;
        h2 += ((t1 >> 24) & 0x3ffffffffffULL) + c; h2 &= 0x3ffffffffffULL;
        const uint64_t lo = h0 | (h1 << 44), hi = (h1 >> 20) | (h2 << 24);
        for (int i = 0; i < 8; i++) { tag[i] = (uint8_t)(lo >> (8 * i)); tag[8 + i] = (uint8_t)(hi >> (8 * i)); }
    }
};

void aead_tag(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
              size_t len, uint8_t tag[16]) {
    uint8_t otk[64];
    chacha_block(key, nonce, 0, otk);
    Poly1305 p(otk);
    if (alen) { p.update(aad, alen); p.pad16(); }
    if (len) { p.update(ct, len); p.pad16(); }
    uint8_t lens[16];
    for (int i = 0; i < 8; i++) { lens[i] = (uint8_t)((uint64_t)alen >> (8 * i)); lens[8 + i] = (uint8_t)((uint64_t)len >> (8 * i)); }
    p.update(lens, 16);
    p.final(tag);
    secure_zero(otk, sizeof otk);
}
}  // namespace

void aead_seal(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* pt,
               size_t len, uint8_t* ct_out, uint8_t tag[16]) {
    chacha_xor(key, nonce, 1, pt, ct_out, len);
    aead_tag(key, nonce, aad, alen, ct_out, len, tag);
}
bool aead_open(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
               size_t len, const uint8_t tag[16], uint8_t* pt_out) {
    uint8_t mine[16];
    aead_tag(key, nonce, aad, alen, ct, len, mine);
    uint8_t diff = 0;
    for (int i = 0; i < 16; i++) diff |= (uint8_t)(mine[i] ^ tag[i]);       // constant time
    secure_zero(mine, sizeof mine);
    if (diff) { if (len) memset(pt_out, 0, len); return false; }
    chacha_xor(key, nonce, 1, ct, pt_out, len);
    return true;
}

// ------------------------------------------------------------------ the reference's payload-AEAD tag
namespace {
// Poly1305 over `m` (a multiple of 16 bytes, as the AEAD always supplies) with 26-bit limbs, followed
// by the reference's tag assembly (S:255-269), which differs from RFC 8439 as described in tf_crypto.h.
void poly1305_turtle(uint8_t tag[16], const uint8_t* m, size_t n, const uint8_t key[32]) {
    const uint64_t M26 = 0x3ffffff;
    const uint64_t r0 = ld32(key) & 0x3ffffff, r1 = (ld32(key + 3) >> 2) & 0x3ffff03, r2 = (ld32(key + 6) >> 4) & 0x3ffc0ff,
                   r3 = (ld32(key + 9) >> 6) & 0x3f03fff, r4 = (ld32(key + 12) >> 8) & 0x00fffff;
    const uint64_t s1 = r1 * 5, s2 = r2 * 5, s3 = r3 * 5, s4 = r4 * 5;
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
    for (size_t off = 0; off < n; off += 16) {
        uint8_t b[16] = {0};
        memcpy(b, m + off, n - off < 16 ? n - off : 16);
        h0 += ld32(b) & M26; h1 += (ld32(b + 3) >> 2) & M26; h2 += (ld32(b + 6) >> 4) & M26;
        h3 += (ld32(b + 9) >> 6) & M26; h4 += (ld32(b + 12) >> 8) | (1ull << 24);
        const uint64_t d0 = h0 * r0 + h1 * s4 + h2 * s3 + h3 * s2 + h4 * s1;
        uint64_t d1 = h0 * r1 + h1 * r0 + h2 * s4 + h3 * s3 + h4 * s2;
        uint64_t d2 = h0 * r2 + h1 * r1 + h2 * r0 + h3 * s4 + h4 * s3;
        uint64_t d3 = h0 * r3 + h1 * r2 + h2 * r1 + h3 * r0 + h4 * s4;
        uint64_t d4 = h0 * r4 + h1 * r3 + h2 * r2 + h3 * r1 + h4 * r0;
        uint64_t c = d0 >> 26; h0 = d0 & M26;
        d1 += c; c = d1 >> 26; h1 = d1 & M26;
        d2 += c; c = d2 >> 26; h2 = d2 & M26;
        d3 += c; c = d3 >> 26; h3 = d3 & M26;
        d4 += c; c = d4 >> 26; h4 = d4 & M26;
        h0 += c * 5; c = h0 >> 26; h0 &= M26; h1 += c;
    }
    uint64_t c = h1 >> 26; h1 &= M26; h2 += c;
    c = h2 >> 26; h2 &= M26; h3 += c;
    c = h3 >> 26; h3 &= M26; h4 += c;
    c = h4 >> 26; h4 &= M26; h0 += c * 5;
    c = h0 >> 26; h0 &= M26; h1 += c;
    uint64_t g0 = h0 + 5; c = g0 >> 26; g0 &= M26;
    uint64_t g1 = h1 + c; c = g1 >> 26; g1 &= M26;
    uint64_t g2 = h2 + c; c = g2 >> 26; g2 &= M26;
    uint64_t g3 = h3 + c; c = g3 >> 26; g3 &= M26;
    const uint64_t g4 = h4 + c - (1ull << 26);
    const uint64_t mask = (g4 >> 63) - 1;
    h0 = (h0 & ~mask) | (g0 & mask); h1 = (h1 & ~mask) | (g1 & mask); h2 = (h2 & ~mask) | (g2 & mask);
    h3 = (h3 & ~mask) | (g3 & mask); h4 = ((h4 & ~mask) | (g4 & mask)) + (1ull << 26);
    // S:261-264: the limb unions are NOT truncated to 32 bits before the carries are taken
    uint64_t f0 = (h0 | (h1 << 26)) + ld32(key + 16);
    uint64_t f1 = ((h1 >> 6) | (h2 << 20)) + ld32(key + 20) + (f0 >> 32); f0 &= 0xffffffff;
    uint64_t f2 = ((h2 >> 12) | (h3 << 14)) + ld32(key + 24) + (f1 >> 32); f1 &= 0xffffffff;
    uint64_t f3 = ((h3 >> 18) | (h4 << 8)) + ld32(key + 28) + (f2 >> 32); f2 &= 0xffffffff; f3 &= 0xffffffff;
    st32(tag, (uint32_t)f0); st32(tag + 4, (uint32_t)f1); st32(tag + 8, (uint32_t)f2); st32(tag + 12, (uint32_t)f3);
}
void aead_tag_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
                     size_t len, uint8_t tag[16]) {
    uint8_t otk[64];
    chacha_block(key, nonce, 0, otk);
    std::vector<uint8_t> mac;          // aad || pad16 || ct || pad16 || le64(alen) || le64(len)   (S:281-288)
    if (aad && alen) { mac.insert(mac.end(), aad, aad + alen); while (mac.size() % 16) mac.push_back(0); }
    if (len) { mac.insert(mac.end(), ct, ct + len); while (mac.size() % 16) mac.push_back(0); }
    for (int i = 0; i < 8; i++) mac.push_back((uint8_t)((uint64_t)alen >> (8 * i)));
    for (int i = 0; i < 8; i++) mac.push_back((uint8_t)((uint64_t)len >> (8 * i)));
    poly1305_turtle(tag, mac.data(), mac.size(), otk);
    secure_zero(mac.data(), mac.size()); secure_zero(otk, sizeof otk);
}
}  // namespace

void aead_seal_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* pt,
                      size_t len, uint8_t* ct_out, uint8_t tag[16]) {
    chacha_xor(key, nonce, 1, pt, ct_out, len);
    aead_tag_turtle(key, nonce, aad, alen, ct_out, len, tag);
}
bool aead_open_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
                      size_t len, const uint8_t tag[16], uint8_t* pt_out) {
    uint8_t mine[16];
    aead_tag_turtle(key, nonce, aad, alen, ct, len, mine);
    uint8_t diff = 0;
    for (int i = 0; i < 16; i++) diff |= (uint8_t)(mine[i] ^ tag[i]);
    secure_zero(mine, sizeof mine);
    if (diff) { if (len) memset(pt_out, 0, len); return false; }
    chacha_xor(key, nonce, 1, ct, pt_out, len);
    return true;
}

// ------------------------------------------------------------------ base64 / random / hex
static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
std::string base64_encode(const uint8_t* d, size_t n) {
    std::string o;
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t v = ((uint32_t)d[i] << 16) | ((i + 1 < n ? (uint32_t)d[i + 1] : 0) << 8) | (i + 2 < n ? d[i + 2] : 0);
        o.push_back(B64[(v >> 18) & 63]); o.push_back(B64[(v >> 12) & 63]);
        o.push_back(i + 1 < n ? B64[(v >> 6) & 63] : '=');
        o.push_back(i + 2 < n ? B64[v & 63] : '=');
    }
    return o;
}
bool base64_decode(const std::string& s, std::vector<uint8_t>& out) {
    out.clear();
    uint32_t acc = 0; int bits = 0; bool padded = false;
    for (char ch : s) {
        if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
        if (ch == '=') { padded = true; continue; }
        if (padded) return false;
        const char* p = strchr(B64, ch);
        if (!p || !ch) return false;
        acc = (acc << 6) | (uint32_t)(p - B64); bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}
bool random_bytes(uint8_t* out, size_t n) {
    size_t got = 0;
    while (got < n) {
        const ssize_t r = getrandom(out + got, n - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got == n) return true;
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return false;
    const bool ok = fread(out, 1, n, f) == n;
    fclose(f);
    return ok;
}
std::string to_hex(const uint8_t* d, size_t n) {
    static const char H[] = "0123456789abcdef";
    std::string o;
    for (size_t i = 0; i < n; i++) { o.push_back(H[d[i] >> 4]); o.push_back(H[d[i] & 15]); }
    return o;
}

}  // namespace tfh
