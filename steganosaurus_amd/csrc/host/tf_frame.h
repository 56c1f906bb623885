// tf_frame.h -- stream framing, ECC and key schedule of the turtlefft wire format
// (replaces steganosaur.cpp:447-508, 550-591, 879-904, 1020-1066).  Host only.
#pragma once
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace tfh {

constexpr size_t HEADER_LEN = 38;      // "FTTG" ver flags salt16 nonce12 clen_be32   (S:886-904)

struct KeyMaterial { std::array<uint8_t, 32> aead_key; std::array<uint8_t, 12> nonce; std::array<uint8_t, 16> salt; };

// PBKDF2(pass,salt,iters) -> HKDF-Extract(no salt) -> HKDF-Expand("fft_turtle:keys", 76) -> [unused32|aead32|nonce12]  (S:556-573)
KeyMaterial derive_keys(const std::string& pass, const std::array<uint8_t, 16>& salt, uint32_t iters);
// HKDF-Extract(salt, master) -> same expand   (S:576-591)
KeyMaterial derive_keys_from_raw(const std::array<uint8_t, 32>& master, const std::array<uint8_t, 16>& salt);

// path_key = SHA256(pass | master_key [|| cover_hash])  (S:1020-1040);  sub = walk|R|G|B  (S:1054-1061)
std::array<uint8_t, 32> path_key_of(const uint8_t* secret, size_t n, const uint8_t* cover_hash32 /*or null*/);
void turtle_subkeys(const std::array<uint8_t, 32>& path_key, uint8_t sub[128]);

std::vector<uint8_t> bits_from_bytes(const std::vector<uint8_t>& bytes);     // MSB first, one byte per bit (S:455-459)
std::vector<uint8_t> bytes_from_bits(const std::vector<uint8_t>& bits);      // S:447-454
std::vector<uint8_t> rep_encode(const std::vector<uint8_t>& bits, int k);    // S:463-467, S:494-500
std::vector<uint8_t> rep_decode(const std::vector<uint8_t>& bits, int k, bool& ok);   // majority, S:468-474, S:501-508

std::vector<uint8_t> header_bytes(const std::array<uint8_t, 16>& salt, const std::array<uint8_t, 12>& nonce, uint32_t clen);

// Rep-3(header) || Rep-7(ciphertext || tag), AAD = header (S:946-995)
std::vector<uint8_t> frame_stream(const KeyMaterial& km, const std::string& secret);

// compute_cover_hash's quantiser (S:433): q = min(7, max(0, floor(log(1+mag)/2)))
std::array<uint8_t, 32> cover_hash_from_mags(const double* mags, size_t n);

}  // namespace tfh
