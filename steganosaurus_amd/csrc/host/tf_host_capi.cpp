// tf_host_capi.cpp -- C entry points over the CLI's host-only pieces (crypto, framing, PNG) so that
// the CPU test-suite can check them against RFC vectors and the reference-made goldens without a GPU.
// Built as steganosaurus_amd/libtfhost.so; not part of the device library.
#include <string.h>

#include "tf_crypto.h"
#include "tf_frame.h"
#include "tf_png.h"

using namespace tfh;

extern "C" {

void tfh_sha256(const uint8_t* d, size_t n, uint8_t out[32]) { auto h = sha256(d, n); memcpy(out, h.data(), 32); }
void tfh_hmac_sha256(const uint8_t* k, size_t kl, const uint8_t* m, size_t ml, uint8_t out[32]) { hmac_sha256(k, kl, m, ml, out); }
void tfh_pbkdf2(const uint8_t* p, size_t pl, const uint8_t* s, size_t sl, uint32_t it, uint8_t* out, size_t n) { pbkdf2_hmac_sha256(p, pl, s, sl, it, out, n); }
void tfh_hkdf_extract(const uint8_t* s, size_t sl, const uint8_t* ikm, size_t il, uint8_t prk[32]) { hkdf_extract(s, sl, ikm, il, prk); }
void tfh_hkdf_expand(const uint8_t prk[32], const uint8_t* info, size_t il, uint8_t* out, size_t n) { hkdf_expand(prk, info, il, out, n); }
void tfh_aead_seal(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t al, const uint8_t* pt, size_t n,
                   uint8_t* ct, uint8_t tag[16]) { aead_seal(key, nonce, aad, al, pt, n, ct, tag); }
int tfh_aead_open(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t al, const uint8_t* ct, size_t n,
                  const uint8_t tag[16], uint8_t* pt) { return aead_open(key, nonce, aad, al, ct, n, tag, pt) ? 1 : 0; }

int tfh_aead_open_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t al, const uint8_t* ct, size_t n,
                         const uint8_t tag[16], uint8_t* pt) { return aead_open_turtle(key, nonce, aad, al, ct, n, tag, pt) ? 1 : 0; }

// Rep-3(header) || Rep-7(ct || tag) for a passphrase and a caller-fixed salt; returns the bit count (0 if cap is too small)
uint64_t tfh_frame_bits(const char* pass, const uint8_t salt16[16], uint32_t iters, const uint8_t* secret, uint32_t slen,
                        uint8_t* bits_out, uint64_t cap) {
    std::array<uint8_t, 16> salt; memcpy(salt.data(), salt16, 16);
    const KeyMaterial km = derive_keys(pass, salt, iters);
    const std::vector<uint8_t> bits = frame_stream(km, std::string((const char*)secret, slen));
    if (bits.size() > cap) return 0;
    memcpy(bits_out, bits.data(), bits.size());
    return bits.size();
}
// inverse: raw bits -> secret; returns length or -1 magic, -2 version, -3 short, -4 auth
int64_t tfh_deframe_bits(const char* pass, uint32_t iters, const uint8_t* bits, uint64_t n, uint8_t* out, uint64_t cap) {
    const size_t hb = HEADER_LEN * 24;
    if (n < hb) return -3;
    bool ok;
    const std::vector<uint8_t> hdr = bytes_from_bits(rep_decode(std::vector<uint8_t>(bits, bits + hb), 3, ok));
    if (memcmp(hdr.data(), "FTTG", 4) != 0) return -1;
    if (hdr[4] != 2) return -2;
    const uint32_t clen = ((uint32_t)hdr[34] << 24) | ((uint32_t)hdr[35] << 16) | ((uint32_t)hdr[36] << 8) | hdr[37];
    const size_t need = ((size_t)clen + 16) * 56;
    if (n < hb + need || clen > cap) return -3;
    const std::vector<uint8_t> rest = bytes_from_bits(rep_decode(std::vector<uint8_t>(bits + hb, bits + hb + need), 7, ok));
    std::array<uint8_t, 16> salt; memcpy(salt.data(), &hdr[6], 16);
    const KeyMaterial km = derive_keys(pass, salt, iters);
    if (!aead_open_turtle(km.aead_key.data(), km.nonce.data(), hdr.data(), HEADER_LEN, rest.data(), clen, rest.data() + clen, out)) return -4;
    return (int64_t)clen;
}
void tfh_turtle_subkeys(const uint8_t* secret, size_t n, uint8_t path_key_out[32], uint8_t sub[128]) {
    const auto pk = path_key_of(secret, n, nullptr);
    memcpy(path_key_out, pk.data(), 32);
    turtle_subkeys(pk, sub);
}

// compute_cover_hash's quantiser + SHA-256 (S:433-443) over magnitudes from tfft_lowfreq_mag, and the path key it feeds
// (S:1020-1040): secret = passphrase bytes or the 32-byte master key; cover_hash32 may be NULL
void tfh_cover_hash_from_mags(const double* mags, size_t n, uint8_t out[32]) { const auto h = cover_hash_from_mags(mags, n); memcpy(out, h.data(), 32); }
void tfh_path_key(const uint8_t* secret, size_t n, const uint8_t* cover_hash32, uint8_t out[32]) { const auto k = path_key_of(secret, n, cover_hash32); memcpy(out, k.data(), 32); }

int tfh_png_write(const char* path, const uint8_t* rgb, int w, int h) { return png_write_rgb8(path, rgb, w, h) ? 0 : -1; }
// two-call protocol: rgb == NULL returns the size through w,h
int tfh_image_read(const char* path, uint8_t* rgb, uint64_t cap, int* w, int* h) {
    std::vector<uint8_t> v;
    if (!load_rgb8(path, v, *w, *h)) return -1;
    if (rgb) { if (v.size() > cap) return -2; memcpy(rgb, v.data(), v.size()); }
    return 0;
}

}  // extern "C"
