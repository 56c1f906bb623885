/* oracle/turtle_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch plain-C (fp64) restatement of the hot path of
 * rickenator/steganosaurus: per-plane 2-D FFT, keyed turtle walk, phase-bit
 * embed/extract.  It is the CHECKER for the HIP path -- only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (steganosaurus_amd/, libturtlefft_hip.so, the turtlefft CLI) never
 * does, and fails loudly when its HIP library is missing.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 * (a) bit-for-bit against oracle/_ref/libtfref.so -- the reference's own
 * translation unit compiled in place by oracle/Makefile -- when that library
 * is present, and (b) against the golden vectors under tests/golden/ that
 * tests/gen_golden.py produced from that same library (the reference ships no
 * test vectors of its own for this path).
 *
 * Each function cites the reference lines it follows as S:<line> of
 * steganosaurus/src/steganosaur.cpp.  Build: see oracle/Makefile
 * (-ffp-contract=off: results must not move with FMA contraction).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double alpha, rmin, rmax, magmin, density, jitter;
    int center, adaptive_alpha;
} orc_params;

/* ------------------------------------------------------------------ SHA-256
 * FIPS 180-4; the reference's copy is S:46-94. */
static const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; size_t fill; } sha_ctx;

static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha_init(sha_ctx* c) {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                                   0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(c->h, iv, sizeof iv); c->len = 0; c->fill = 0;
}
static void sha_update(sha_ctx* c, const uint8_t* d, size_t n) {
    c->len += n;
    while (n) {
        size_t k = 64 - c->fill; if (k > n) k = n;
        memcpy(c->buf + c->fill, d, k); c->fill += k; d += k; n -= k;
        if (c->fill == 64) { sha_block(c->h, c->buf); c->fill = 0; }
    }
}
static void sha_final(sha_ctx* c, uint8_t out[32]) {
    uint64_t bits = c->len * 8;
    uint8_t pad = 0x80; sha_update(c, &pad, 1);
    uint8_t z = 0; while (c->fill != 56) sha_update(c, &z, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (8 * (7 - i)));
    sha_update(c, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = c->h[i] >> 24; out[4 * i + 1] = c->h[i] >> 16; out[4 * i + 2] = c->h[i] >> 8; out[4 * i + 3] = c->h[i]; }
}
void orc_sha256(const uint8_t* d, size_t n, uint8_t out[32]) { sha_ctx c; sha_init(&c); sha_update(&c, d, n); sha_final(&c, out); }

/* HMAC-SHA-256 (RFC 2104; S:96-110) and HKDF-Expand (RFC 5869; S:135-147). */
static void hmac256(const uint8_t* key, size_t klen, const uint8_t* msg, size_t mlen, uint8_t out[32]) {
    uint8_t k0[64] = {0}, ipad[64], opad[64], inner[32];
    if (klen > 64) orc_sha256(key, klen, k0); else memcpy(k0, key, klen);
    for (int i = 0; i < 64; i++) { ipad[i] = k0[i] ^ 0x36; opad[i] = k0[i] ^ 0x5c; }
    sha_ctx c; sha_init(&c); sha_update(&c, ipad, 64); sha_update(&c, msg, mlen); sha_final(&c, inner);
    sha_init(&c); sha_update(&c, opad, 64); sha_update(&c, inner, 32); sha_final(&c, out);
}
static void hkdf_expand(const uint8_t prk[32], const uint8_t* info, size_t ilen, uint8_t* out, size_t L) {
    uint8_t T[32], msg[32 + 64 + 1]; size_t tlen = 0, pos = 0; uint8_t ctr = 1;
    while (pos < L) {
        memcpy(msg, T, tlen); memcpy(msg + tlen, info, ilen); msg[tlen + ilen] = ctr;
        hmac256(prk, 32, msg, tlen + ilen + 1, T); tlen = 32;
        size_t need = L - pos < 32 ? L - pos : 32;
        memcpy(out + pos, T, need); pos += need; ctr++;
    }
}
/* S:1054-1061: walk | R | G | B sub-keys; path_key is used directly as PRK. */
void orc_subkeys(const uint8_t path_key[32], uint8_t sub[128]) {
    hkdf_expand(path_key, (const uint8_t*)"turtle_keys", 11, sub, 128);
}

/* -------------------------------------------------------------- keystream KS
 * S:665-695.  Block k = SHA256(key || 0xAA || le32(k)). */
typedef struct { uint8_t key[32], state[32]; size_t pos; uint32_t ctr; int bitpool, bits; } ks_t;
static void ks_init(ks_t* k, const uint8_t key[32]) { memcpy(k->key, key, 32); memset(k->state, 0, 32); k->pos = 32; k->ctr = 0; k->bitpool = 0; k->bits = 0; }
static uint8_t ks_byte(ks_t* k) {                                   /* S:673-684 */
    if (k->pos >= 32) {
        uint8_t m[37]; memcpy(m, k->key, 32); m[32] = 0xAA;
        m[33] = (uint8_t)k->ctr; m[34] = (uint8_t)(k->ctr >> 8); m[35] = (uint8_t)(k->ctr >> 16); m[36] = (uint8_t)(k->ctr >> 24);
        orc_sha256(m, 37, k->state); k->pos = 0; k->ctr++;
    }
    return k->state[k->pos++];
}
static int ks_opcode3(ks_t* k) {                                    /* S:685 */
    /* the reference shifts a signed int left without masking; only the low
     * (bits) bits are ever read back, so an unsigned pool is equivalent */
    while (k->bits < 3) { k->bitpool = (int)(((unsigned)k->bitpool << 8) | ks_byte(k)); k->bits += 8; }
    int op = (k->bitpool >> (k->bits - 3)) & 7; k->bits -= 3; return op;
}
static int ks_hit_density(ks_t* k, double density) {                /* S:686-689 */
    /* (uint8_t)floor(density*256.0): out-of-range conversion wraps mod 256 on
     * x86-64 (density >= 1.0 gives 0: the reference then never hits) */
    uint8_t thr = (uint8_t)(long long)floor(density * 256.0);
    return ks_byte(k) < thr;
}
static double ks_jitter(ks_t* k, double maxj) {                     /* S:690-694 */
    /* (next_byte()<<8)|next_byte(): g++ evaluates the left operand first
     * (checked against oracle/_ref in tests/test_oracle.py) */
    int hi = ks_byte(k); int lo = ks_byte(k);
    int16_t r = (int16_t)((hi << 8) | lo);
    double u = r / 32768.0;
    return u * maxj;
}
void orc_ks_bytes(const uint8_t key[32], size_t n, uint8_t* out) { ks_t k; ks_init(&k, key); for (size_t i = 0; i < n; i++) out[i] = ks_byte(&k); }
void orc_ks_opcodes(const uint8_t key[32], size_t n, uint8_t* out) { ks_t k; ks_init(&k, key); for (size_t i = 0; i < n; i++) out[i] = (uint8_t)ks_opcode3(&k); }

/* ------------------------------------------------------------------- FFT
 * S:341-358: in-place radix-2 DIT, bit reversal, twiddle by recurrence,
 * exp(+2*pi*i/len) forward, exp(-...) inverse, inverse divides by n.
 * a = n interleaved (re,im) doubles. */
void orc_fft1d(double* a, int n, int inverse) {
    for (size_t i = 1, j = 0; i < (size_t)n; i++) {                 /* S:343-345 */
        size_t bit = (size_t)n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double tr = a[2 * i], ti = a[2 * i + 1]; a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1]; a[2 * j] = tr; a[2 * j + 1] = ti; }
    }
    for (size_t len = 2; len <= (size_t)n; len <<= 1) {             /* S:346-356 */
        double ang = 2 * M_PI / len * (inverse ? -1 : 1);
        double lr = cos(ang), li = sin(ang);
        for (size_t i = 0; i < (size_t)n; i += len) {
            double wr = 1, wi = 0;
            for (size_t j = 0; j < len / 2; j++) {
                double* u = a + 2 * (i + j); double* v = a + 2 * (i + j + len / 2);
                double vr = v[0] * wr - v[1] * wi, vi = v[0] * wi + v[1] * wr;
                double ur = u[0], ui = u[1];
                u[0] = ur + vr; u[1] = ui + vi; v[0] = ur - vr; v[1] = ui - vi;
                double nr = wr * lr - wi * li, ni = wr * li + wi * lr; wr = nr; wi = ni;
            }
        }
    }
    if (inverse) for (int i = 0; i < 2 * n; i++) a[i] /= (double)n;  /* S:357 */
}
/* S:359-366: all rows, then all columns.  a = H*W interleaved complex. */
void orc_fft2d(double* a, int H, int W, int inverse) {
    for (int y = 0; y < H; y++) orc_fft1d(a + 2 * (size_t)y * W, W, inverse);
    double* col = (double*)malloc(sizeof(double) * 2 * (size_t)H);
    for (int x = 0; x < W; x++) {
        for (int y = 0; y < H; y++) { col[2 * y] = a[2 * ((size_t)y * W + x)]; col[2 * y + 1] = a[2 * ((size_t)y * W + x) + 1]; }
        orc_fft1d(col, H, inverse);
        for (int y = 0; y < H; y++) { a[2 * ((size_t)y * W + x)] = col[2 * y]; a[2 * ((size_t)y * W + x) + 1] = col[2 * y + 1]; }
    }
    free(col);
}

/* -------------------------------------------------------------- geometry */
int orc_next_pow2(int v) { size_t p = 1; while (p < (size_t)v) p <<= 1; return (int)p; }       /* S:369 */
static void conj_idx(int y, int x, int H, int W, int* cy, int* cx) {                           /* S:370-372 */
    int yy = (y == 0) ? 0 : (H - y), xx = (x == 0) ? 0 : (W - x); *cy = yy % H; *cx = xx % W;
}
static int on_axis(int y, int x, int H, int W) {                                               /* S:698-700 */
    return (y == 0 || x == 0 || (H % 2 == 0 && y == H / 2) || (W % 2 == 0 && x == W / 2));
}
static int imin(int a, int b) { return a < b ? a : b; }

/* ------------------------------------------------------------- plane I/O
 * S:383-398: de-interleave, optional (-1)^(x+y), zero-pad to pow2, forward. */
static double* forward3(const uint8_t* rgb, int W, int H, int center, int* PWo, int* PHo) {
    int PW = orc_next_pow2(W), PH = orc_next_pow2(H);
    size_t P = (size_t)PW * PH;
    double* F = (double*)calloc(3 * P * 2, sizeof(double));
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                double v = rgb[3 * ((size_t)y * W + x) + p];                                   /* S:385 */
                if (center && ((x + y) & 1)) v *= -1.0;                                        /* S:392 */
                F[2 * (p * P + (size_t)y * PW + x)] = v;                                       /* S:396 */
            }
    for (int p = 0; p < 3; p++) orc_fft2d(F + 2 * p * P, PH, PW, 0);                           /* S:921 */
    *PWo = PW; *PHo = PH;
    return F;
}

/* S:404-409: element at sorted index n/2 of |F| (upper median).  cabs==hypot. */
static double select_kth(double* v, size_t n, size_t k) {
    /* Hoare quickselect: value at sorted index k (== std::nth_element's v[k]) */
    long lo = 0, hi = (long)n - 1;
    while (lo < hi) {
        double piv = v[lo + (hi - lo) / 2];
        long i = lo - 1, j = hi + 1;
        for (;;) {
            do i++; while (v[i] < piv);
            do j--; while (v[j] > piv);
            if (i >= j) break;
            double t = v[i]; v[i] = v[j]; v[j] = t;
        }
        if ((long)k <= j) hi = j; else lo = j + 1;
    }
    return v[k];
}
double orc_median_abs(const double* plane, int PH, int PW) {
    size_t n = (size_t)PH * PW;
    double* m = (double*)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) m[i] = hypot(plane[2 * i], plane[2 * i + 1]);
    double r = select_kth(m, n, n / 2);
    free(m); return r;
}

void orc_forward_rgb8(const uint8_t* rgb, int W, int H, int center, double* out, double* med) {
    int PW, PH; double* F = forward3(rgb, W, H, center, &PW, &PH);
    size_t P = (size_t)PW * PH;
    if (med) for (int p = 0; p < 3; p++) med[p] = orc_median_abs(F + 2 * p * P, PH, PW);
    if (out) memcpy(out, F, 3 * P * 2 * sizeof(double));
    free(F);
}

/* S:415-444 compute_cover_hash, as do_embed / do_extract call it (S:1021-1033, S:1157-1169: on the planes after
 * apply_center): forward transform of each plane, |F[y][x]| for y, x < region = min(8, min(PH,PW)/8) (S:429),
 * q = min(7, max(0, floor(log(1+mag)/2))) (S:433), SHA-256 over the 3*region^2 bytes in plane, y, x order.
 * mags_out / q_out optional.  Returns region. */
int orc_cover_hash(const uint8_t* rgb, int W, int H, int center, uint8_t hash_out[32], double* mags_out, uint8_t* q_out) {
    int PW, PH; double* F = forward3(rgb, W, H, center, &PW, &PH);
    size_t P = (size_t)PW * PH;
    int region = imin(8, imin(PH, PW) / 8);
    uint8_t q[192];
    size_t n = 0;
    for (int p = 0; p < 3; p++)
        for (int y = 0; y < region; y++)
            for (int x = 0; x < region; x++) {
                const double* z = F + 2 * (p * P + (size_t)y * PW + x);
                double mag = hypot(z[0], z[1]);
                double v = floor(log(1.0 + mag) / 2.0);
                if (v < 0.0) v = 0.0;
                if (v > 7.0) v = 7.0;
                if (mags_out) mags_out[n] = mag;
                q[n++] = (uint8_t)v;
            }
    if (q_out) memcpy(q_out, q, n);
    orc_sha256(q, n, hash_out);
    free(F);
    return region;
}

/* S:998-1008: per-plane count over the annulus with magnitude floor, c/2. */
static uint64_t capacity_plane(const double* F, int PH, int PW, double rmin, double rmax, double t) {
    uint64_t c = 0; int mn = imin(PH, PW);
    for (int y = 0; y < PH; y++)
        for (int x = 0; x < PW; x++) {
            if (on_axis(y, x, PH, PW)) continue;
            if (y == 0 && x == 0) continue;
            double r = hypot((double)y, (double)x);
            if (r < rmin * mn || r > rmax * mn) continue;
            size_t i = (size_t)y * PW + x;
            if (hypot(F[2 * i], F[2 * i + 1]) < t) continue;
            int cy, cx; conj_idx(y, x, PH, PW, &cy, &cx);
            if (!(cy == y && cx == x)) c++;
        }
    return c / 2;
}
uint64_t orc_capacity_rgb8(const uint8_t* rgb, int W, int H, const orc_params* P, double med_out[3]) {
    int PW, PH; double* F = forward3(rgb, W, H, P->center, &PW, &PH);
    size_t Pn = (size_t)PW * PH; uint64_t usable = 0;
    for (int p = 0; p < 3; p++) {
        double m = orc_median_abs(F + 2 * p * Pn, PH, PW);
        if (med_out) med_out[p] = m;
        usable += capacity_plane(F + 2 * p * Pn, PH, PW, P->rmin, P->rmax, P->magmin * m);      /* S:923 */
    }
    free(F); return usable;
}

/* ------------------------------------------------------------ Turtle walk
 * S:749-810. */
typedef struct { int y, x, plane, H, W; ks_t* ks; uint8_t* visited; double rmin, rmax; } turtle_t;

static void turtle_init(turtle_t* T, int H, int W, ks_t* ks, double rmin, double rmax) {
    T->H = H; T->W = W; T->ks = ks; T->rmin = rmin; T->rmax = rmax;
    T->visited = (uint8_t*)calloc((size_t)3 * H * W, 1);                                       /* S:759 */
    char pre[64]; int n = snprintf(pre, sizeof pre, "seed:%dx%d|key:", H, W);                  /* S:764-766 */
    sha_ctx c; uint8_t h[32]; sha_init(&c); sha_update(&c, (const uint8_t*)pre, (size_t)n); sha_update(&c, ks->key, 32); sha_final(&c, h);
    uint64_t s = 0; for (int i = 0; i < 8; i++) s = (s << 8) | h[i];                           /* S:768 */
    T->y = (int)((s >> 0) % (uint64_t)H); T->x = (int)((s >> 16) % (uint64_t)W); T->plane = (int)((s >> 32) % 3);  /* S:769 */
}
static int annulus_ok(const turtle_t* T, int yy, int xx) {                                     /* S:771-774 */
    double r = hypot((double)yy, (double)xx); int mn = imin(T->H, T->W);
    return (r >= T->rmin * mn && r <= T->rmax * mn);
}
/* S:778-804.  Returns 0, or -1 when `budget` opcode steps pass without an
 * acceptable bin (the reference would spin forever). */
static int turtle_advance(turtle_t* T, uint64_t budget) {
    const int H = T->H, W = T->W;
    for (uint64_t it = 0; it < budget; it++) {
        int op = ks_opcode3(T->ks);
        switch (op) {
            case 0: T->plane = (T->plane + 1) % 3; break;
            case 1: T->x = (T->x + 1) % W; break;
            case 2: T->y = (T->y + 1) % H; break;
            case 3: T->x = (T->x - 1 + W) % W; break;
            case 4: T->y = (T->y - 1 + H) % H; break;
            case 5: T->x = (T->x + 1) % W; T->y = (T->y + 1) % H; break;
            case 6: T->x = (T->x - 1 + W) % W; T->y = (T->y + 1) % H; break;
            default: break;
        }
        int y = T->y, x = T->x;
        if (on_axis(y, x, H, W)) continue;
        if (y == 0 && x == 0) continue;
        if (T->visited[((size_t)T->plane * H + y) * W + x]) continue;
        if (!annulus_ok(T, y, x)) continue;
        int cy, cx; conj_idx(y, x, H, W, &cy, &cx);
        if (T->visited[((size_t)T->plane * H + cy) * W + cx]) continue;
        return 0;
    }
    return -1;
}
static void turtle_mark(turtle_t* T) {                                                         /* S:805-809 */
    T->visited[((size_t)T->plane * T->H + T->y) * T->W + T->x] = 1;
    int cy, cx; conj_idx(T->y, T->x, T->H, T->W, &cy, &cx);
    T->visited[((size_t)T->plane * T->H + cy) * T->W + cx] = 1;
}
/* advance + density gate: the inner while(true) of S:1076-1081 / S:1206 */
static int turtle_next(turtle_t* T, double density, uint64_t* skipped) {
    uint64_t budget = 64ull * 3 * (uint64_t)T->H * T->W + 4096;
    for (;;) {
        if (turtle_advance(T, budget)) return -1;
        if (ks_hit_density(T->ks, density)) return 0;
        turtle_mark(T); if (skipped) (*skipped)++;
    }
}

int orc_walk(const uint8_t key_walk[32], int PH, int PW, double rmin, double rmax, double density,
             uint64_t n, int32_t* bins, uint64_t* skipped, uint32_t* ks_ctr, int32_t start[3]) {
    ks_t ks; ks_init(&ks, key_walk); turtle_t T; turtle_init(&T, PH, PW, &ks, rmin, rmax);
    if (start) { start[0] = T.plane; start[1] = T.y; start[2] = T.x; }
    uint64_t sk = 0; int rc = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (turtle_next(&T, density, &sk)) { rc = -1; break; }
        bins[3 * i] = T.plane; bins[3 * i + 1] = T.y; bins[3 * i + 2] = T.x;
        turtle_mark(&T);
    }
    if (skipped) *skipped = sk;
    if (ks_ctr) *ks_ctr = ks.ctr;
    free(T.visited); return rc;
}

/* --------------------------------------------------------- phase write/read */
static double adaptive_alpha(double base, double mag, double med, int on) {                    /* S:704-710 */
    if (!on) return base;
    double scale = fmin(2.0, fmax(0.5, mag / fmax(1e-12, med)));
    return base * scale;
}
static void write_bit(double* F, int PH, int PW, int y, int x, int bit, double base_alpha, double jit, ks_t* ks,
                      double med, int adaptive) {                                              /* S:712-732 */
    size_t i = (size_t)y * PW + x;
    double mag = fmax(1e-12, hypot(F[2 * i], F[2 * i + 1]));
    double alpha = adaptive_alpha(base_alpha, mag, med, adaptive);
    double target = bit ? +alpha : -alpha;
    double j = ks_jitter(ks, jit);
    double theta = target + j;
    double nr = mag * cos(theta), ni = mag * sin(theta);       /* std::polar */
    F[2 * i] = nr; F[2 * i + 1] = ni;
    int cy, cx; conj_idx(y, x, PH, PW, &cy, &cx);
    if (!(cy == y && cx == x)) { size_t c = (size_t)cy * PW + cx; F[2 * c] = nr; F[2 * c + 1] = -ni; }
    else { F[2 * i] = mag; F[2 * i + 1] = 0.0; }
}
static double ang_diff(double a, double b) { double d = fmod(a - b + M_PI, 2 * M_PI); if (d < 0) d += 2 * M_PI; return fabs(d - M_PI); }
static int read_bit(const double* F, int PW, int y, int x, double base_alpha, double joff, double med, int adaptive) {  /* S:734-746 */
    size_t i = (size_t)y * PW + x;
    double th = atan2(F[2 * i + 1], F[2 * i]);
    double mag = fmax(1e-12, hypot(F[2 * i], F[2 * i + 1]));
    double alpha = adaptive_alpha(base_alpha, mag, med, adaptive);
    double dpos = ang_diff(th, joff + alpha), dneg = ang_diff(th, joff - alpha);
    return (dpos <= dneg) ? 1 : 0;
}

/* ------------------------------------------------- embed / extract (signal)
 * S:912-923, S:1054-1103 without PNG and crypto. */
int orc_embed_rgb8(const uint8_t* rgb, int W, int H, const orc_params* P, const uint8_t path_key[32],
                   const uint8_t* bits, uint64_t n_bits, uint8_t* out_rgb, double* spec_out, int32_t* bins_out) {
    int PW, PH; double* F = forward3(rgb, W, H, P->center, &PW, &PH);
    size_t Pn = (size_t)PW * PH; double med[3];
    for (int p = 0; p < 3; p++) med[p] = orc_median_abs(F + 2 * p * Pn, PH, PW);
    uint8_t sub[128]; orc_subkeys(path_key, sub);
    ks_t ksw, ksp[3]; ks_init(&ksw, sub); for (int p = 0; p < 3; p++) ks_init(&ksp[p], sub + 32 * (p + 1));
    turtle_t T; turtle_init(&T, PH, PW, &ksw, P->rmin, P->rmax);
    int rc = 0;
    for (uint64_t i = 0; i < n_bits; i++) {                                                    /* S:1074-1097 */
        if (turtle_next(&T, P->density, NULL)) { rc = -1; break; }
        if (bins_out) { bins_out[3 * i] = T.plane; bins_out[3 * i + 1] = T.y; bins_out[3 * i + 2] = T.x; }
        write_bit(F + 2 * (size_t)T.plane * Pn, PH, PW, T.y, T.x, bits[i], P->alpha, P->jitter, &ksp[T.plane], med[T.plane], P->adaptive_alpha);
        turtle_mark(&T);
    }
    free(T.visited);
    if (spec_out) memcpy(spec_out, F, 3 * Pn * 2 * sizeof(double));
    if (rc == 0 && out_rgb) {
        for (int p = 0; p < 3; p++) orc_fft2d(F + 2 * p * Pn, PH, PW, 1);                      /* S:1100 */
        for (int p = 0; p < 3; p++)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    double v = F[2 * (p * Pn + (size_t)y * PW + x)];                           /* S:401 */
                    if (P->center && ((x + y) & 1)) v *= -1.0;                                 /* S:1102 */
                    out_rgb[3 * ((size_t)y * W + x) + p] = (uint8_t)fmax(0.0, fmin(255.0, round(v)));  /* S:389 */
                }
    }
    free(F); return rc;
}

int orc_extract_bits(const uint8_t* rgb, int W, int H, const orc_params* P, const uint8_t path_key[32],
                     uint64_t n_bits, uint8_t* bits_out) {                                     /* S:1116-1132, S:1185-1220 */
    int PW, PH; double* F = forward3(rgb, W, H, P->center, &PW, &PH);
    size_t Pn = (size_t)PW * PH; double med[3] = {0, 0, 0};
    if (P->adaptive_alpha) for (int p = 0; p < 3; p++) med[p] = orc_median_abs(F + 2 * p * Pn, PH, PW);
    uint8_t sub[128]; orc_subkeys(path_key, sub);
    ks_t ksw, ksp[3]; ks_init(&ksw, sub); for (int p = 0; p < 3; p++) ks_init(&ksp[p], sub + 32 * (p + 1));
    turtle_t T; turtle_init(&T, PH, PW, &ksw, P->rmin, P->rmax);
    int rc = 0;
    for (uint64_t i = 0; i < n_bits; i++) {
        if (turtle_next(&T, P->density, NULL)) { rc = -1; break; }
        double j = ks_jitter(&ksp[T.plane], P->jitter);                                        /* S:1208 */
        bits_out[i] = (uint8_t)read_bit(F + 2 * (size_t)T.plane * Pn, PW, T.y, T.x, P->alpha, j, med[T.plane], P->adaptive_alpha);
        turtle_mark(&T);
    }
    free(T.visited); free(F); return rc;
}

/* Raw bits of a spectrum given explicit bins (used to check the HIP gather on
 * spectra the HIP path itself produced).  spec = 3 planes PH*PW complex fp64. */
void orc_read_bins(const double* spec, int PH, int PW, const int32_t* bins, const double* jit, uint64_t n,
                   double alpha, int adaptive, const double med[3], uint8_t* bits_out) {
    size_t Pn = (size_t)PW * PH;
    for (uint64_t i = 0; i < n; i++) {
        int p = bins[3 * i];
        bits_out[i] = (uint8_t)read_bit(spec + 2 * p * Pn, PW, bins[3 * i + 1], bins[3 * i + 2], alpha,
                                        jit ? jit[i] : 0.0, med ? med[p] : 0.0, adaptive);
    }
}
