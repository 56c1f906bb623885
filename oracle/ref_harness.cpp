// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the *unmodified* reference translation unit where it lies under
// /root/reference (found through -I on the command line, see oracle/Makefile)
// and exposes its file-static hot-path functions through a small C ABI so that
// tests/gen_golden.py and the parity tests can drive the real reference code.
// Outputs go to oracle/_ref/ only.  No reference source text lives in this repo:
// this file only *calls* the reference's functions (cited below as S:line of
// steganosaurus/src/steganosaur.cpp).
//
// The composition in ref_embed_rgb8 / ref_extract_bits re-states the order of
// calls of do_embed (S:907-1109) and do_extract (S:1112-1312) minus PNG I/O and
// crypto, because those two functions are monolithic (file in, file out, random
// salt).  The compiled CLI (oracle/_ref/turtlefft) is used by
// tests/gen_golden.py to confirm that this composition interoperates with the
// real do_embed/do_extract in both directions.

#define main turtlefft_reference_main
#include "steganosaur.cpp"
#undef main

#include <cstdint>
#include <cstring>

extern "C" {

struct ref_params {
    double alpha, rmin, rmax, magmin, density, jitter;
    int center, adaptive_alpha;
};

int ref_next_pow2(int v) { return (int)next_pow2((size_t)v); }   // S:369

void ref_sha256(const uint8_t* d, size_t n, uint8_t out[32]) {    // S:64
    auto h = sha256::hash(d, n);
    memcpy(out, h.data(), 32);
}

// S:1054-1061: sub-keys walk|R|G|B = HKDF-Expand(PRK=path_key, "turtle_keys", 128)
void ref_subkeys(const uint8_t path_key[32], uint8_t sub[128]) {
    const uint8_t info[] = "turtle_keys";
    sha256::hkdf_sha256_expand(path_key, info, sizeof(info) - 1, sub, 128);
}

void ref_ks_bytes(const uint8_t key[32], size_t n, uint8_t* out) {  // S:665-684
    array<uint8_t, 32> k; memcpy(k.data(), key, 32);
    KS ks(k);
    for (size_t i = 0; i < n; i++) out[i] = ks.next_byte();
}

void ref_ks_opcodes(const uint8_t key[32], size_t n, uint8_t* out) {  // S:685
    array<uint8_t, 32> k; memcpy(k.data(), key, 32);
    KS ks(k);
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)ks.next_opcode3();
}

// 1-D FFT, interleaved re/im doubles.  S:341-358
void ref_fft1d(double* data, int n, int inverse) {
    vector<complex<double>> a(n);
    for (int i = 0; i < n; i++) a[i] = complex<double>(data[2 * i], data[2 * i + 1]);
    fft1d(a, inverse != 0);
    for (int i = 0; i < n; i++) { data[2 * i] = a[i].real(); data[2 * i + 1] = a[i].imag(); }
}

// 2-D FFT on a row-major H x W interleaved complex array.  S:359-366
void ref_fft2d(double* data, int H, int W, int inverse) {
    vector<vector<complex<double>>> A(H, vector<complex<double>>(W));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            A[y][x] = complex<double>(data[2 * ((size_t)y * W + x)], data[2 * ((size_t)y * W + x) + 1]);
    fft2d(A, inverse != 0);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            data[2 * ((size_t)y * W + x)] = A[y][x].real();
            data[2 * ((size_t)y * W + x) + 1] = A[y][x].imag();
        }
}

typedef vector<vector<complex<double>>> Plane;

static void forward3(const uint8_t* rgb, int W, int H, bool center, Plane F[3], int& PW, int& PH) {
    vector<double> R, G, B;
    to_planes_u8(rgb, W, H, 3, R, G, B);                                   // S:912
    apply_center(R, W, H, center); apply_center(G, W, H, center); apply_center(B, W, H, center);  // S:914
    F[0] = pad_to_fft(R, W, H, PW, PH); F[1] = pad_to_fft(G, W, H, PW, PH); F[2] = pad_to_fft(B, W, H, PW, PH);  // S:917
    fft2d(F[0], false); fft2d(F[1], false); fft2d(F[2], false);            // S:921
}

// Forward spectra of an RGB8 image: out = 3 planes of PH*PW interleaved complex
// doubles.  med[3] = median_abs (S:922).  Either output may be NULL.
void ref_forward_rgb8(const uint8_t* rgb, int W, int H, int center, double* out, double* med) {
    Plane F[3]; int PW, PH;
    forward3(rgb, W, H, center != 0, F, PW, PH);
    for (int p = 0; p < 3; p++) {
        if (med) med[p] = median_abs(F[p]);
        if (out) {
            double* o = out + (size_t)p * PH * PW * 2;
            for (int y = 0; y < PH; y++)
                for (int x = 0; x < PW; x++) {
                    o[2 * ((size_t)y * PW + x)] = F[p][y][x].real();
                    o[2 * ((size_t)y * PW + x) + 1] = F[p][y][x].imag();
                }
        }
    }
}

// Capacity estimate, the count_plane lambda of S:998-1008 re-stated over the
// reference's own helpers (the lambda itself is local to do_embed).
static size_t capacity_plane(const Plane& F, int PH, int PW, double rmin, double rmax, double t) {
    size_t c = 0;
    for (int y = 0; y < PH; y++)
        for (int x = 0; x < PW; x++) {
            if (on_axis(y, x, PH, PW)) continue;
            if (y == 0 && x == 0) continue;
            double r = hypot_idx(y, x);
            if (r < rmin * min(PH, PW) || r > rmax * min(PH, PW)) continue;
            if (abs(F[y][x]) < t) continue;
            auto [cy, cx] = conj_idx(y, x, PH, PW);
            if (!(cy == y && cx == x)) c++;
        }
    return c / 2;
}

uint64_t ref_capacity_rgb8(const uint8_t* rgb, int W, int H, const ref_params* P, double med_out[3]) {
    Plane F[3]; int PW, PH;
    forward3(rgb, W, H, P->center != 0, F, PW, PH);
    uint64_t usable = 0;
    for (int p = 0; p < 3; p++) {
        double m = median_abs(F[p]);
        if (med_out) med_out[p] = m;
        usable += capacity_plane(F[p], PH, PW, P->rmin, P->rmax, P->magmin * m);   // S:923, S:1008
    }
    return usable;
}

// The keyed walk with the density gate, exactly the loop of S:1074-1081 /
// S:1206 with the write/read removed.  bins = n triples (plane,y,x).
// Returns 0.  start[3] = (plane,y,x) straight after Turtle construction.
int ref_walk(const uint8_t key_walk[32], int PH, int PW, double rmin, double rmax, double density,
             uint64_t n, int32_t* bins, uint64_t* skipped, uint32_t* ks_ctr, int32_t start[3]) {
    array<uint8_t, 32> kw; memcpy(kw.data(), key_walk, 32);
    KS ks_walk(kw);
    array<KS*, 3> planes = {nullptr, nullptr, nullptr};
    Turtle T(PH, PW, &ks_walk, planes, rmin, rmax, nullptr, vector<double>{0, 0, 0});
    if (start) { start[0] = T.plane; start[1] = T.y; start[2] = T.x; }
    uint64_t sk = 0;
    for (uint64_t i = 0; i < n; i++) {
        while (true) {
            T.advance_to_valid();
            if (ks_walk.hit_density(density)) break;
            T.mark_here(); sk++;
        }
        bins[3 * i + 0] = T.plane; bins[3 * i + 1] = T.y; bins[3 * i + 2] = T.x;
        T.mark_here();
    }
    if (skipped) *skipped = sk;
    if (ks_ctr) *ks_ctr = ks_walk.ctr;
    return 0;
}

struct SubKeys { array<uint8_t, 32> walk, r, g, b; };
static SubKeys subkeys_of(const uint8_t path_key[32]) {
    uint8_t sub[128]; ref_subkeys(path_key, sub);
    SubKeys k;
    memcpy(k.walk.data(), sub, 32); memcpy(k.r.data(), sub + 32, 32);
    memcpy(k.g.data(), sub + 64, 32); memcpy(k.b.data(), sub + 96, 32);
    return k;
}

// Signal path of do_embed (S:912-923, S:1015, S:1054-1103) for a given bit
// vector and path key.  out_rgb = W*H*3 stego bytes.  spec_out (optional) =
// the three spectra after the embed loop, before the inverse transform.
// bins_out (optional) = n_bits triples.
int ref_embed_rgb8(const uint8_t* rgb, int W, int H, const ref_params* P, const uint8_t path_key[32],
                   const uint8_t* bits, uint64_t n_bits, uint8_t* out_rgb, double* spec_out, int32_t* bins_out) {
    Plane F[3]; int PW, PH;
    forward3(rgb, W, H, P->center != 0, F, PW, PH);
    double med[3] = {median_abs(F[0]), median_abs(F[1]), median_abs(F[2])};
    vector<double> thr = {P->magmin * med[0], P->magmin * med[1], P->magmin * med[2]};
    vector<vector<vector<complex<double>>>> F3 = {F[0], F[1], F[2]};        // S:1015
    SubKeys k = subkeys_of(path_key);
    KS ks_walk(k.walk), ks_r(k.r), ks_g(k.g), ks_b(k.b);
    array<KS*, 3> ks_planes = {&ks_r, &ks_g, &ks_b};
    vector<double> median_mags = {med[0], med[1], med[2]};
    Turtle T(PH, PW, &ks_walk, ks_planes, P->rmin, P->rmax, &F3, thr);      // S:1071
    for (uint64_t i = 0; i < n_bits; i++) {                                 // S:1074-1097
        while (true) {
            T.advance_to_valid();
            if (ks_walk.hit_density(P->density)) break;
            T.mark_here();
        }
        if (bins_out) { bins_out[3 * i] = T.plane; bins_out[3 * i + 1] = T.y; bins_out[3 * i + 2] = T.x; }
        write_bit_on_bin(F3[T.plane], T.y, T.x, bits[i], P->alpha, P->jitter,
                         *ks_planes[T.plane], median_mags[T.plane], P->adaptive_alpha != 0);
        T.mark_here();
    }
    if (spec_out) {
        for (int p = 0; p < 3; p++) {
            double* o = spec_out + (size_t)p * PH * PW * 2;
            for (int y = 0; y < PH; y++)
                for (int x = 0; x < PW; x++) {
                    o[2 * ((size_t)y * PW + x)] = F3[p][y][x].real();
                    o[2 * ((size_t)y * PW + x) + 1] = F3[p][y][x].imag();
                }
        }
    }
    fft2d(F3[0], true); fft2d(F3[1], true); fft2d(F3[2], true);             // S:1100
    auto R2 = ifft_crop(F3[0], W, H), G2 = ifft_crop(F3[1], W, H), B2 = ifft_crop(F3[2], W, H);
    apply_center(R2, W, H, P->center); apply_center(G2, W, H, P->center); apply_center(B2, W, H, P->center);
    vector<uint8_t> out; from_planes_u8(R2, G2, B2, W, H, out);             // S:1103
    memcpy(out_rgb, out.data(), out.size());
    return 0;
}

// Signal path of do_extract (S:1116-1132, S:1185-1220): the first n_bits raw
// (pre-ECC) bits along the walk.
int ref_extract_bits(const uint8_t* rgb, int W, int H, const ref_params* P, const uint8_t path_key[32],
                     uint64_t n_bits, uint8_t* bits_out) {
    Plane F[3]; int PW, PH;
    forward3(rgb, W, H, P->center != 0, F, PW, PH);
    double med[3] = {median_abs(F[0]), median_abs(F[1]), median_abs(F[2])};
    vector<double> thr = {P->magmin * med[0], P->magmin * med[1], P->magmin * med[2]};
    vector<vector<vector<complex<double>>>> F3 = {F[0], F[1], F[2]};
    SubKeys k = subkeys_of(path_key);
    KS ks_walk(k.walk), ks_r(k.r), ks_g(k.g), ks_b(k.b);
    array<KS*, 3> ks_planes = {&ks_r, &ks_g, &ks_b};
    vector<double> median_mags = {med[0], med[1], med[2]};
    Turtle T(PH, PW, &ks_walk, ks_planes, P->rmin, P->rmax, &F3, thr);
    for (uint64_t i = 0; i < n_bits; i++) {                                 // S:1205-1220
        while (true) { T.advance_to_valid(); if (ks_walk.hit_density(P->density)) break; T.mark_here(); }
        double j = ks_planes[T.plane]->jitter(P->jitter);
        bits_out[i] = (uint8_t)read_bit_from_bin(F3[T.plane], T.y, T.x, P->alpha, j,
                                                 median_mags[T.plane], P->adaptive_alpha != 0);
        T.mark_here();
    }
    return 0;
}

// Framing of do_embed (S:942-995) with a caller-fixed salt: returns the number
// of stream bits written to bits_out (one byte per bit), or 0 if cap too small.
uint64_t ref_frame_bits(const char* pass, const uint8_t salt16[16], uint32_t iters,
                        const uint8_t* secret, uint32_t slen, uint8_t* bits_out, uint64_t cap) {
    array<uint8_t, 16> salt; memcpy(salt.data(), salt16, 16);
    KeyMaterial km = derive_keys(string(pass), salt, iters);               // S:942
    Header Hdr; Hdr.salt = km.salt; Hdr.nonce = km.nonce; Hdr.clen = slen; // S:946
    vector<uint8_t> header_bytes = Hdr.to_bytes();
    vector<uint8_t> ct(secret, secret + slen);
    array<uint8_t, 16> tag{};
    chacha_poly::chacha20_poly1305_seal(km.aead_key.data(), km.nonce.data(), header_bytes.data(),
                                        header_bytes.size(), ct.data(), ct.size(), tag.data());   // S:970
    auto header_rep3 = rep3_encode_bits(bits_from_bytes(header_bytes));    // S:986-987
    vector<uint8_t> payload_bytes(ct.begin(), ct.end());
    payload_bytes.insert(payload_bytes.end(), tag.begin(), tag.end());
    auto payload_rep7 = rep7_encode_bits(bits_from_bytes(payload_bytes));  // S:990-991
    uint64_t n = header_rep3.size() + payload_rep7.size();
    if (n > cap) return 0;
    memcpy(bits_out, header_rep3.data(), header_rep3.size());
    memcpy(bits_out + header_rep3.size(), payload_rep7.data(), payload_rep7.size());
    return n;
}

// De-framing of do_extract (S:1223-1311) from a raw bit vector.  Returns the
// secret length (>=0) or a negative code: -1 magic, -2 version, -3 short, -4 auth.
int64_t ref_deframe_bits(const char* pass, uint32_t iters, const uint8_t* bits, uint64_t n_bits,
                         uint8_t* secret_out, uint64_t cap) {
    size_t hb = Header::fixed_len() * 8 * 3;
    if (n_bits < hb) return -3;
    bool ok = true;
    vector<uint8_t> h3(bits, bits + hb);
    auto hdr_bytes = bytes_from_bits(rep3_decode_bits(h3, ok));
    if (!(hdr_bytes[0] == 'F' && hdr_bytes[1] == 'T' && hdr_bytes[2] == 'T' && hdr_bytes[3] == 'G')) return -1;
    if (hdr_bytes[4] != 2) return -2;
    uint32_t clen = u32be_read(&hdr_bytes[34]);
    size_t need = ((size_t)clen + 16) * 8 * 7;
    if (n_bits < hb + need) return -3;
    vector<uint8_t> r7(bits + hb, bits + hb + need);
    auto rest = bytes_from_bits(rep7_decode_bits(r7, ok));
    array<uint8_t, 16> salt; memcpy(salt.data(), &hdr_bytes[6], 16);
    KeyMaterial km = derive_keys(string(pass), salt, iters);
    vector<uint8_t> ct(rest.begin(), rest.begin() + clen);
    array<uint8_t, 16> tag; memcpy(tag.data(), rest.data() + clen, 16);
    vector<uint8_t> aad(hdr_bytes.begin(), hdr_bytes.begin() + Header::fixed_len());
    if (!chacha_poly::chacha20_poly1305_open(km.aead_key.data(), km.nonce.data(), aad.data(), aad.size(),
                                             ct.data(), ct.size(), tag.data())) return -4;
    if (clen > cap) return -3;
    memcpy(secret_out, ct.data(), clen);
    return (int64_t)clen;
}

// compute_cover_hash (S:415-444) as do_embed / do_extract call it (S:1021-1033, S:1157-1169): on the
// de-interleaved planes AFTER apply_center (S:914).  hash_out = the reference's own 32 bytes.  mags_out
// (optional, 3*region*region doubles) = the magnitudes its lambda quantises, recomputed here with the
// reference's pad_to_fft / fft2d / abs so that fixtures can carry them; q_out (optional) = their quantised
// bytes.  Returns region = min(8, min(PH,PW)/8) (S:429).
int ref_cover_hash(const uint8_t* rgb, int W, int H, int center, uint8_t hash_out[32], double* mags_out, uint8_t* q_out) {
    vector<double> R, G, B;
    to_planes_u8(rgb, W, H, 3, R, G, B);
    apply_center(R, W, H, center != 0); apply_center(G, W, H, center != 0); apply_center(B, W, H, center != 0);
    auto h = compute_cover_hash(R, G, B, W, H);
    memcpy(hash_out, h.data(), 32);
    int PW = 0, PH = 0, region = 0;
    const vector<double>* planes[3] = {&R, &G, &B};
    for (int p = 0; p < 3; p++) {
        auto F = pad_to_fft(*planes[p], W, H, PW, PH);
        fft2d(F, false);
        region = min(8, min(PH, PW) / 8);
        for (int y = 0; y < region; y++)
            for (int x = 0; x < region; x++) {
                double mag = abs(F[y][x]);
                size_t i = ((size_t)p * region + y) * region + x;
                if (mags_out) mags_out[i] = mag;
                if (q_out) q_out[i] = (uint8_t)min(7.0, max(0.0, floor(log(1.0 + mag) / 2.0)));
            }
    }
    return region;
}

// PNG I/O through the reference's vendored stb (S:909, S:1104) so fixtures and
// the CLI agree on the container.
int ref_png_write(const char* path, const uint8_t* rgb, int W, int H) {
    return stbi_write_png(path, W, H, 3, rgb, W * 3) ? 0 : -1;
}
int ref_png_read(const char* path, uint8_t* rgb_out, uint64_t cap, int* W, int* H) {
    int comp; stbi_uc* img = stbi_load(path, W, H, &comp, 3);
    if (!img) return -1;
    size_t n = (size_t)(*W) * (*H) * 3;
    int rc = 0;
    if (rgb_out) { if (n <= cap) memcpy(rgb_out, img, n); else rc = -2; }
    stbi_image_free(img);
    return rc;
}

}  // extern "C"
