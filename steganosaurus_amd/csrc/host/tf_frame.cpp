#include "tf_frame.h"

#include <math.h>
#include <string.h>

#include "tf_crypto.h"

namespace tfh {

static KeyMaterial expand_keys(uint8_t prk[32], const std::array<uint8_t, 16>& salt) {
    uint8_t out[76];
    hkdf_expand(prk, (const uint8_t*)"fft_turtle:keys", 15, out, sizeof out);
    KeyMaterial km;
    memcpy(km.aead_key.data(), out + 32, 32);       // bytes 0..31 ("path_key") are never used by the reference (S:565)
    memcpy(km.nonce.data(), out + 64, 12);
    km.salt = salt;
    secure_zero(out, sizeof out); secure_zero(prk, 32);
    return km;
}
KeyMaterial derive_keys(const std::string& pass, const std::array<uint8_t, 16>& salt, uint32_t iters) {
    uint8_t dk[32], prk[32];
    pbkdf2_hmac_sha256((const uint8_t*)pass.data(), pass.size(), salt.data(), salt.size(), iters, dk, 32);
    hkdf_extract(nullptr, 0, dk, 32, prk);
    secure_zero(dk, 32);
    return expand_keys(prk, salt);
}
KeyMaterial derive_keys_from_raw(const std::array<uint8_t, 32>& master, const std::array<uint8_t, 16>& salt) {
    uint8_t prk[32];
    hkdf_extract(salt.data(), salt.size(), master.data(), master.size(), prk);
    return expand_keys(prk, salt);
}

std::array<uint8_t, 32> path_key_of(const uint8_t* secret, size_t n, const uint8_t* cover_hash32) {
    Sha256 s; s.update(secret, n);
    if (cover_hash32) s.update(cover_hash32, 32);
    std::array<uint8_t, 32> out; s.final(out.data());
    return out;
}
void turtle_subkeys(const std::array<uint8_t, 32>& path_key, uint8_t sub[128]) {
    hkdf_expand(path_key.data(), (const uint8_t*)"turtle_keys", 11, sub, 128);
}

std::vector<uint8_t> bits_from_bytes(const std::vector<uint8_t>& bytes) {
    std::vector<uint8_t> bits; bits.reserve(bytes.size() * 8);
    for (uint8_t b : bytes) for (int i = 7; i >= 0; --i) bits.push_back((b >> i) & 1);
    return bits;
}
std::vector<uint8_t> bytes_from_bits(const std::vector<uint8_t>& bits) {
    std::vector<uint8_t> out; out.reserve(bits.size() / 8 + 1);
    for (size_t i = 0; i < bits.size(); i += 8) {
        uint8_t v = 0;
        for (int j = 0; j < 8; j++) v = (uint8_t)((v << 1) | ((i + j < bits.size()) ? bits[i + j] : 0));
        out.push_back(v);
    }
    return out;
}
std::vector<uint8_t> rep_encode(const std::vector<uint8_t>& bits, int k) {
    std::vector<uint8_t> out; out.reserve(bits.size() * k);
    for (uint8_t b : bits) for (int i = 0; i < k; i++) out.push_back(b);
    return out;
}
std::vector<uint8_t> rep_decode(const std::vector<uint8_t>& bits, int k, bool& ok) {
    ok = (bits.size() % k) == 0;
    std::vector<uint8_t> out; out.reserve(bits.size() / k);
    for (size_t i = 0; i + k <= bits.size(); i += k) {
        int s = 0;
        for (int j = 0; j < k; j++) s += bits[i + j];
        out.push_back(s >= (k + 1) / 2 ? 1 : 0);
    }
    return out;
}

std::vector<uint8_t> header_bytes(const std::array<uint8_t, 16>& salt, const std::array<uint8_t, 12>& nonce, uint32_t clen) {
    std::vector<uint8_t> b = {'F', 'T', 'T', 'G', 2, 0};
    b.insert(b.end(), salt.begin(), salt.end());
    b.insert(b.end(), nonce.begin(), nonce.end());
    b.push_back((uint8_t)(clen >> 24)); b.push_back((uint8_t)(clen >> 16)); b.push_back((uint8_t)(clen >> 8)); b.push_back((uint8_t)clen);
    return b;
}

std::vector<uint8_t> frame_stream(const KeyMaterial& km, const std::string& secret) {
    const std::vector<uint8_t> hdr = header_bytes(km.salt, km.nonce, (uint32_t)secret.size());
    std::vector<uint8_t> payload(secret.size() + 16);
    aead_seal_turtle(km.aead_key.data(), km.nonce.data(), hdr.data(), hdr.size(), (const uint8_t*)secret.data(), secret.size(),
              payload.data(), payload.data() + secret.size());
    std::vector<uint8_t> bits = rep_encode(bits_from_bytes(hdr), 3);
    const std::vector<uint8_t> p7 = rep_encode(bits_from_bytes(payload), 7);
    bits.insert(bits.end(), p7.begin(), p7.end());
    return bits;
}

std::array<uint8_t, 32> cover_hash_from_mags(const double* mags, size_t n) {
    std::vector<uint8_t> q(n);
    for (size_t i = 0; i < n; i++) q[i] = (uint8_t)fmin(7.0, fmax(0.0, floor(log(1.0 + mags[i]) / 2.0)));
    return sha256(q.data(), q.size());
}

}  // namespace tfh
