// tf_crypto.h -- host crypto of the turtlefft CLI, written from the RFCs (not from the reference):
//   SHA-256 (FIPS 180-4), HMAC (RFC 2104), PBKDF2 (RFC 8018), HKDF (RFC 5869),
//   ChaCha20-Poly1305 AEAD (RFC 8439), base64, OS CSPRNG.
// These stay on the host (north_star); they replace steganosaur.cpp:46-323 and the parts of
// crypto/crypto_utils.h the CLI uses.  Wire compatibility with the reference is pinned by
// tests/golden/kat.json (frames produced by the reference's own code).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace tfh {

void secure_zero(void* p, size_t n);

struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len;
    size_t fill;
    Sha256();
    void update(const void* data, size_t n);
    void final(uint8_t out[32]);
};
std::array<uint8_t, 32> sha256(const void* data, size_t n);

void hmac_sha256(const uint8_t* key, size_t klen, const uint8_t* msg, size_t mlen, uint8_t out[32]);
void pbkdf2_hmac_sha256(const uint8_t* pass, size_t plen, const uint8_t* salt, size_t slen, uint32_t iters,
                        uint8_t* out, size_t dklen);
void hkdf_extract(const uint8_t* salt, size_t slen, const uint8_t* ikm, size_t ilen, uint8_t prk[32]);
void hkdf_expand(const uint8_t prk[32], const uint8_t* info, size_t ilen, uint8_t* out, size_t L);

// RFC 8439.  seal: ct_out may alias pt.  open: returns false (and leaves pt_out zeroed) on a bad tag.
void aead_seal(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* pt,
               size_t len, uint8_t* ct_out, uint8_t tag[16]);
bool aead_open(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
               size_t len, const uint8_t tag[16], uint8_t* pt_out);

// The payload AEAD of the turtlefft wire format.  The reference's in-TU Poly1305
// (steganosaur.cpp:192-270) assembles the final tag from its 26-bit limbs in 64-bit
// arithmetic WITHOUT truncating (h0 | h1<<26) etc. to 32 bits before propagating carries
// (S:261-264), so limb bits are counted twice and the tag is NOT the RFC 8439 tag.  Stego
// images are only interoperable with the reference if that exact value is reproduced.  The
// reference's library AEAD (crypto/chacha20poly1305.cpp:100-190, used for key wrapping) has the
// same tag assembly, so the CLI uses these two everywhere; aead_seal/aead_open above are the
// RFC-conformant primitive (tested against the RFC 8439 vector) and are not used on the wire.
void aead_seal_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* pt,
                      size_t len, uint8_t* ct_out, uint8_t tag[16]);
bool aead_open_turtle(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t alen, const uint8_t* ct,
                      size_t len, const uint8_t tag[16], uint8_t* pt_out);

std::string base64_encode(const uint8_t* d, size_t n);
bool base64_decode(const std::string& s, std::vector<uint8_t>& out);   // false on a malformed string
bool random_bytes(uint8_t* out, size_t n);
std::string to_hex(const uint8_t* d, size_t n);

}  // namespace tfh
