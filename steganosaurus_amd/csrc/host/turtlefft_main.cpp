// turtlefft_main.cpp -- the `turtlefft embed | extract | gen-key` command line of
// rickenator/steganosaurus, re-implemented as a host driver around the MI355X library
// (include/turtlefft_hip.h).  Command grammar, defaults (struct Params, S:375-381),
// success/error messages and exit codes follow steganosaur.cpp:813-877, 907-1426 so that
// scripts written for the reference keep working; the signal path (planes, FFT, medians,
// capacity, phase embed/read, inverse) runs on the GPU through the C ABI, the keyed walk,
// framing, ECC and crypto run on the host.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/turtlefft_hip.h"
#include "tf_crypto.h"
#include "tf_frame.h"
#include "tf_png.h"

using namespace tfh;

namespace {

struct Params {     // S:375-381
    double alpha = 0.50, rmin = 0.05, rmax = 0.45, magmin = 0.01, density = 0.7, jitter = 0.0;
    bool center = false;
    uint32_t pbkdf2_iter = 600000;
    bool adaptive_alpha = false;
    bool cover_dependent_path = false;
};
struct Args {       // S:839-845
    std::string mode, inPath, outPath, secret, pass, keyBase64, keyOutPath, wrapPass;
    Params P;
};

void usage() {
    fprintf(stderr,
            "Usage:\n"
            "  turtlefft gen-key [--key-out FILE] [--wrap-pass PW]\n"
            "      Generate a 256-bit master key; print base64 and fingerprint; optionally export it\n"
            "      (wrapped with ChaCha20-Poly1305 under PW when --wrap-pass is given).\n"
            "  turtlefft embed   --in host.png --out stego.png --secret TEXT (--pass PW | --key KEY_BASE64)\n"
            "      [--alpha 0.5 --jitter 0 --density 0.7 --rmin 0.05 --rmax 0.45 --magmin 0.01 --center 0]\n"
            "      [--pbkdf2_iter 600000 --adaptive_alpha 0 --cover_dependent_path 0 --wrap-pass PW]\n"
            "  turtlefft extract --in stego.png (--pass PW | --key KEY_BASE64)\n"
            "      [same tuning options as embed]\n"
            "  The 2-D FFT / phase embedding runs on an AMD MI355X through libturtlefft_hip.so.\n");
}

// ---- command line: one table row per option (the grammar, defaults and failure modes are the reference's, S:813-877: every
// option takes exactly one value, unknown options print `Unknown arg: X` + usage, flags are "1" / "true")
enum class OptKind { Text, Real, Flag, Count };
struct OptSpec {
    const char* name;
    OptKind kind;
    std::string Args::*text;          // Text
    double Params::*real;             // Real
    bool Params::*flag;               // Flag
    uint32_t Params::*count;          // Count
};
const OptSpec kOptions[] = {
    {"--in", OptKind::Text, &Args::inPath, nullptr, nullptr, nullptr},
    {"--out", OptKind::Text, &Args::outPath, nullptr, nullptr, nullptr},
    {"--secret", OptKind::Text, &Args::secret, nullptr, nullptr, nullptr},
    {"--pass", OptKind::Text, &Args::pass, nullptr, nullptr, nullptr},
    {"--key", OptKind::Text, &Args::keyBase64, nullptr, nullptr, nullptr},
    {"--key-out", OptKind::Text, &Args::keyOutPath, nullptr, nullptr, nullptr},
    {"--wrap-pass", OptKind::Text, &Args::wrapPass, nullptr, nullptr, nullptr},
    {"--alpha", OptKind::Real, nullptr, &Params::alpha, nullptr, nullptr},
    {"--jitter", OptKind::Real, nullptr, &Params::jitter, nullptr, nullptr},
    {"--density", OptKind::Real, nullptr, &Params::density, nullptr, nullptr},
    {"--rmin", OptKind::Real, nullptr, &Params::rmin, nullptr, nullptr},
    {"--rmax", OptKind::Real, nullptr, &Params::rmax, nullptr, nullptr},
    {"--magmin", OptKind::Real, nullptr, &Params::magmin, nullptr, nullptr},
    {"--center", OptKind::Flag, nullptr, nullptr, &Params::center, nullptr},
    {"--adaptive_alpha", OptKind::Flag, nullptr, nullptr, &Params::adaptive_alpha, nullptr},
    {"--cover_dependent_path", OptKind::Flag, nullptr, nullptr, &Params::cover_dependent_path, nullptr},
    {"--pbkdf2_iter", OptKind::Count, nullptr, nullptr, nullptr, &Params::pbkdf2_iter},
};

const OptSpec* find_option(const char* name) {
    for (const OptSpec& o : kOptions)
        if (strcmp(o.name, name) == 0) return &o;
    return nullptr;
}

bool parse_args(int argc, char** argv, Args& A) {
    if (argc < 2) return false;
    A.mode = argv[1];
    int i = 2;
    while (i < argc) {
        const OptSpec* o = find_option(argv[i]);
        if (!o) { fprintf(stderr, "Unknown arg: %s\n", argv[i]); return false; }
        const std::string value = (i + 1 < argc) ? argv[i + 1] : "";      // a trailing option without a value reads as empty
        i += 2;
        switch (o->kind) {
            case OptKind::Text: A.*(o->text) = value; break;
            case OptKind::Real: A.P.*(o->real) = std::stod(value); break;               // throws on garbage: main() prints usage
            case OptKind::Flag: A.P.*(o->flag) = (value == "1" || value == "true"); break;
            case OptKind::Count: A.P.*(o->count) = (uint32_t)std::stoul(value); break;
        }
    }
    // what each mode cannot do without
    const bool has_secret_source = !A.pass.empty() || !A.keyBase64.empty();
    if (A.mode == "gen-key") return true;
    if (A.mode == "extract") return !A.inPath.empty() && has_secret_source;
    if (A.mode == "embed") return !A.inPath.empty() && has_secret_source && !A.outPath.empty() && !A.secret.empty();
    return false;
}

[[noreturn]] void die(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void die(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
    fputc('\n', stderr);
    exit(1);
}
void tf(int rc, const char* what) {
    if (rc != TFFT_OK) die("%s: %s", what, tfft_strerror(rc));
}

// 80-byte wrapped key: "TFKW" salt16 nonce12 ct32 tag16, or a raw 32-byte key, base64 (S:594-662)
bool decode_or_unwrap_key(const std::string& key_data, const std::string& unwrap_pass, uint32_t iters,
                          std::array<uint8_t, 32>& key_out) {
    std::vector<uint8_t> d;
    if (!base64_decode(key_data, d)) return false;
    if (d.empty() && !key_data.empty()) return false;
    if (d.size() == 80 && memcmp(d.data(), "TFKW", 4) == 0) {
        if (unwrap_pass.empty()) { fprintf(stderr, "Key is wrapped but no unwrap passphrase provided\n"); return false; }
        uint8_t derived[44];
        pbkdf2_hmac_sha256((const uint8_t*)unwrap_pass.data(), unwrap_pass.size(), d.data() + 4, 16, iters, derived, 44);
        const bool ok = aead_open_turtle(derived, d.data() + 20, nullptr, 0, d.data() + 32, 32, d.data() + 64, key_out.data());
        secure_zero(derived, sizeof derived);
        return ok;
    }
    if (d.size() == 32) { memcpy(key_out.data(), d.data(), 32); return true; }
    return false;
}

struct Spectrum {        // a resident image on the device + everything derived from the key
    tfft_ctx* ctx = nullptr;
    int W = 0, H = 0, PW = 0, PH = 0;
    double med[3] = {0, 0, 0};
    std::array<uint8_t, 32> path_key;
    uint8_t sub[128];
    ~Spectrum() { if (ctx) tfft_destroy(ctx); secure_zero(sub, sizeof sub); }
};

// S:909-923 / S:1114-1131 and S:1020-1066: load, forward transform, medians, path key, sub-keys
void open_image(const Args& A, Spectrum& S, const std::vector<uint8_t>& rgb, bool using_raw_key,
                const std::array<uint8_t, 32>& master_key) {
    int dev = 0;
    if (const char* e = getenv("TURTLEFFT_DEVICE")) dev = atoi(e);
    tf(tfft_create(dev, S.W, S.H, 1, &S.ctx), "turtlefft: cannot open the GPU context");
    tf(tfft_forward_rgb8(S.ctx, 0, rgb.data(), S.W, S.H, A.P.center, &S.PW, &S.PH), "forward transform");
    tf(tfft_medians(S.ctx, 0, S.med), "medians");
    const uint8_t* sec = using_raw_key ? master_key.data() : (const uint8_t*)A.pass.data();
    const size_t sec_n = using_raw_key ? 32 : A.pass.size();
    if (A.P.cover_dependent_path) {      // S:415-444 on the resident spectrum: 8x8 low-frequency magnitudes
        const int region = std::min(8, std::min(S.PH, S.PW) / 8);
        std::vector<double> mags((size_t)3 * region * region);
        if (region > 0) tf(tfft_lowfreq_mag(S.ctx, 0, region, mags.data()), "cover hash");
        const auto ch = cover_hash_from_mags(mags.data(), mags.size());
        S.path_key = path_key_of(sec, sec_n, ch.data());
    } else {
        S.path_key = path_key_of(sec, sec_n, nullptr);
    }
    turtle_subkeys(S.path_key, S.sub);
}

void do_embed(const Args& A) {      // S:907-1109
    const bool using_raw_key = !A.keyBase64.empty();
    std::array<uint8_t, 32> master_key{};
    Spectrum S;
    std::vector<uint8_t> rgb;      // the reference loads the image before it looks at the key (S:909 vs S:935)
    if (!load_rgb8(A.inPath, rgb, S.W, S.H)) die("Failed to load %s", A.inPath.c_str());
    if (using_raw_key && !decode_or_unwrap_key(A.keyBase64, A.wrapPass, A.P.pbkdf2_iter, master_key))
        die("Failed to decode/unwrap key from --key argument");
    open_image(A, S, rgb, using_raw_key, master_key);
    std::vector<uint8_t>().swap(rgb);

    std::array<uint8_t, 16> salt{};
    { std::random_device rd; for (auto& b : salt) b = (uint8_t)rd(); }                  // S:927-929
    const KeyMaterial km = using_raw_key ? derive_keys_from_raw(master_key, salt) : derive_keys(A.pass, salt, A.P.pbkdf2_iter);
    secure_zero(master_key.data(), master_key.size());
    const std::vector<uint8_t> bits = frame_stream(km, A.secret);

    const double thr[3] = {A.P.magmin * S.med[0], A.P.magmin * S.med[1], A.P.magmin * S.med[2]};   // S:923
    uint64_t usable = 0;
    tf(tfft_capacity(S.ctx, 0, A.P.rmin, A.P.rmax, thr, &usable), "capacity");
    if (bits.size() > usable)
        die("Message too large. Need %zu bits (after ECC), capacity ~%zu bits.", bits.size(), (size_t)usable);

    tfft_walk* wk = nullptr;
    tf(tfft_walk_create(S.sub, S.PH, S.PW, A.P.rmin, A.P.rmax, A.P.density, &wk), "walk");
    std::vector<tfft_bin> bins(bits.size());
    tf(tfft_walk_next(wk, bins.size(), bins.data(), nullptr), "walk");
    tfft_walk_destroy(wk);
    std::vector<float> jit;
    if (A.P.jitter != 0.0) {
        jit.resize(bins.size());
        tf(tfft_walk_jitter(S.sub + 32, bins.data(), bins.size(), A.P.jitter, jit.data()), "jitter");
    }
    tf(tfft_embed_bins(S.ctx, 0, bins.data(), bits.data(), jit.empty() ? nullptr : jit.data(), bins.size(), A.P.alpha,
                       A.P.adaptive_alpha, S.med), "embed");
    std::vector<uint8_t> out((size_t)S.W * S.H * 3);
    tf(tfft_inverse_rgb8(S.ctx, 0, out.data()), "inverse transform");
    if (!png_write_rgb8(A.outPath, out.data(), S.W, S.H)) die("PNG write failed: %s", A.outPath.c_str());
    fprintf(stdout, "Embedded %zu bits into %s (payload %u bytes, ver=2, salt/nonce in header)\n", bits.size(),
            A.outPath.c_str(), (unsigned)A.secret.size());
}

void do_extract(const Args& A) {    // S:1112-1312
    const bool using_raw_key = !A.keyBase64.empty();
    std::array<uint8_t, 32> master_key{};
    Spectrum S;
    std::vector<uint8_t> rgb;
    if (!load_rgb8(A.inPath, rgb, S.W, S.H)) die("Failed to load %s", A.inPath.c_str());
    if (using_raw_key && !decode_or_unwrap_key(A.keyBase64, A.wrapPass, A.P.pbkdf2_iter, master_key))
        die("Failed to decode/unwrap key from --key argument");
    open_image(A, S, rgb, using_raw_key, master_key);
    std::vector<uint8_t>().swap(rgb);

    tfft_walk* wk = nullptr;
    tf(tfft_walk_create(S.sub, S.PH, S.PW, A.P.rmin, A.P.rmax, A.P.density, &wk), "walk");
    uint64_t walked = 0;
    auto read_bits = [&](size_t n) {      // the read_next_bit loop, n positions further along the SAME walk (S:1205-1226)
        std::vector<tfft_bin> bins(n);
        const int rc = tfft_walk_next(wk, n, bins.data(), nullptr);
        if (rc == TFFT_E_EXHAUSTED) die("Payload truncated after ECC decode.");   // the reference would spin forever here
        tf(rc, "walk");
        std::vector<float> jit;
        if (A.P.jitter != 0.0) {
            // the per-plane jitter streams continue across calls: regenerate from the start and keep the tail
            std::vector<tfft_bin> all;   // (only paid when --jitter is used)
            tfft_walk* w2 = nullptr;
            tf(tfft_walk_create(S.sub, S.PH, S.PW, A.P.rmin, A.P.rmax, A.P.density, &w2), "walk");
            all.resize(walked + n);
            tf(tfft_walk_next(w2, all.size(), all.data(), nullptr), "walk");
            tfft_walk_destroy(w2);
            std::vector<float> ja(all.size());
            tf(tfft_walk_jitter(S.sub + 32, all.data(), all.size(), A.P.jitter, ja.data()), "jitter");
            jit.assign(ja.begin() + (ptrdiff_t)walked, ja.end());
        }
        walked += n;
        std::vector<uint8_t> bits(n);
        tf(tfft_read_bins(S.ctx, 0, bins.data(), jit.empty() ? nullptr : jit.data(), n, A.P.alpha, A.P.adaptive_alpha, S.med,
                          bits.data()), "read");
        return bits;
    };

    bool ok = true;
    const std::vector<uint8_t> hdr_bits = rep_decode(read_bits(HEADER_LEN * 8 * 3), 3, ok);
    if (!ok) die("Header ECC length mismatch.");
    const std::vector<uint8_t> hdr = bytes_from_bits(hdr_bits);
    if (hdr.size() < HEADER_LEN) die("Header truncated.");
    if (!(hdr[0] == 'F' && hdr[1] == 'T' && hdr[2] == 'T' && hdr[3] == 'G')) die("Magic not found.");
    if (hdr[4] != 2) die("Unsupported version (%u).", hdr[4]);
    const uint32_t clen = ((uint32_t)hdr[34] << 24) | ((uint32_t)hdr[35] << 16) | ((uint32_t)hdr[36] << 8) | hdr[37];
    const size_t rest_bytes = (size_t)clen + 16;
    // a corrupt header can claim up to 4 GiB: bound the request by what the annulus can hold at all
    // (the reference allocates and walks blindly)
    if (rest_bytes * 56 > (size_t)3 * S.PH * S.PW) die("Payload truncated after ECC decode.");
    const std::vector<uint8_t> pay_bits = rep_decode(read_bits(rest_bytes * 8 * 7), 7, ok);
    if (!ok) die("Payload rep7 decode failed.");
    const std::vector<uint8_t> rest = bytes_from_bits(pay_bits);
    if (rest.size() < rest_bytes) die("Payload truncated after ECC decode.");
    tfft_walk_destroy(wk);

    std::array<uint8_t, 16> salt; memcpy(salt.data(), &hdr[6], 16);
    const KeyMaterial km = using_raw_key ? derive_keys_from_raw(master_key, salt) : derive_keys(A.pass, salt, A.P.pbkdf2_iter);
    secure_zero(master_key.data(), master_key.size());
    std::string secret(clen, '\0');
    // authenticated with the RECEIVED header bytes as AAD and the DERIVED nonce (S:1299-1307)
    if (!aead_open_turtle(km.aead_key.data(), km.nonce.data(), hdr.data(), HEADER_LEN, rest.data(), clen, rest.data() + clen,
                   (uint8_t*)&secret[0]))
        die("Auth failed (wrong pass or data corrupted).");
    printf("%s\n", secret.c_str());
}

void do_gen_key(const Args& A) {    // S:1315-1416
    std::array<uint8_t, 32> master{};
    if (!random_bytes(master.data(), 32)) die("Failed to generate random key (CSPRNG error)");
    const auto fp = sha256(master.data(), 32);
    const std::string b64 = base64_encode(master.data(), 32);
    printf("Generated 256-bit master key:\n  Base64: %s\n  Fingerprint: %s\n", b64.c_str(), to_hex(fp.data(), 8).c_str());
    if (!A.keyOutPath.empty()) {
        std::string text;
        if (!A.wrapPass.empty()) {
            uint8_t salt[16], derived[44];
            if (!random_bytes(salt, 16)) die("Failed to generate salt");
            pbkdf2_hmac_sha256((const uint8_t*)A.wrapPass.data(), A.wrapPass.size(), salt, 16, A.P.pbkdf2_iter, derived, 44);
            std::vector<uint8_t> blob = {'T', 'F', 'K', 'W'};
            blob.insert(blob.end(), salt, salt + 16);
            blob.insert(blob.end(), derived + 32, derived + 44);
            blob.resize(80);
            aead_seal_turtle(derived, derived + 32, nullptr, 0, master.data(), 32, &blob[32], &blob[64]);
            secure_zero(derived, sizeof derived);
            text = base64_encode(blob.data(), blob.size()) + "\n";
            printf("  Wrapped with passphrase and exported to: %s\n", A.keyOutPath.c_str());
        } else {
            text = b64 + "\n";
            printf("  Exported (unencrypted) to: %s\n", A.keyOutPath.c_str());
        }
        FILE* f = fopen(A.keyOutPath.c_str(), "wb");
        if (!f) die("Failed to open %s for writing", A.keyOutPath.c_str());
        const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
        fclose(f);
        if (!ok) die("Failed to write key file");
    }
    secure_zero(master.data(), 32);
}

}  // namespace

int main(int argc, char** argv) {
    Args A;
    try {
        if (!parse_args(argc, argv, A)) { usage(); return 1; }
    } catch (const std::exception&) {      // the reference lets std::stod throw on a missing/garbled value (uncaught -> abort)
        usage();
        return 1;
    }
    if (A.mode == "gen-key") do_gen_key(A);
    else if (A.mode == "embed") do_embed(A);
    else do_extract(A);
    return 0;
}
