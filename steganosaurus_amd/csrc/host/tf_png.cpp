#include "tf_png.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <exception>

namespace tfh {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the per-scanline filters of one (sub)image in place; in = (1+stride)*rows bytes
bool unfilter(uint8_t* in, size_t rows, size_t stride, size_t bpp) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t* line = in + y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pr;
            switch (ft) {
                case 0: pr = 0; break;
                case 1: pr = a; break;
                case 2: pr = b; break;
                case 3: pr = (a + b) >> 1; break;
                case 4: pr = paeth(a, b, c); break;
                default: return false;
            }
            cur[i] = (uint8_t)(cur[i] + pr);
        }
        prev = cur;
    }
    return true;
}

}  // namespace

bool png_decode_rgb8(const uint8_t* d, size_t n, std::vector<uint8_t>& rgb, int& w, int& h) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 || memcmp(d, sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos);
        if (pos + 12 + (size_t)len > n) return false;
        const uint8_t* type = d + pos + 4; const uint8_t* body = d + pos + 8;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return false;
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24)) return false;
    int chans;
    switch (ctype) { case 0: chans = 1; break; case 2: chans = 3; break; case 3: chans = 1; break; case 4: chans = 2; break; case 6: chans = 4; break; default: return false; }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && depth == 16) return false;
    const size_t bits_pp = (size_t)chans * depth;
    const size_t bpp = (bits_pp + 7) / 8;
    // pass geometry (Adam7 or a single pass)
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    size_t total = 0;
    for (int p = 0; p < npass; p++) {
        const size_t pw = interlace ? (W > (uint32_t)xs[p] ? (W - xs[p] + dx[p] - 1) / dx[p] : 0) : W;
        const size_t ph = interlace ? (H > (uint32_t)ys[p] ? (H - ys[p] + dy[p] - 1) / dy[p] : 0) : H;
        if (pw && ph) total += ph * (1 + (pw * bits_pp + 7) / 8);
    }
    // a deflate stream expands at most ~1032x: an IHDR that promises more than the IDAT bytes can hold is a forged
    // (or truncated) file -- refuse it before sizing any buffer from its dimensions; and keep W*H*3 inside what this
    // tool's transforms handle anyway (TFFT_MAX_DIM^2 pixels)
    if (total > idat.size() * 1040 + 4096 || (uint64_t)W * H > (uint64_t)16384 * 16384) return false;
    std::vector<uint8_t> raw(total);
    uLongf dl = (uLongf)total;
    if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK || dl != total) return false;
    rgb.assign((size_t)W * H * 3, 0);
    size_t off = 0;
    for (int p = 0; p < npass; p++) {
        const size_t pw = interlace ? (W > (uint32_t)xs[p] ? (W - xs[p] + dx[p] - 1) / dx[p] : 0) : W;
        const size_t ph = interlace ? (H > (uint32_t)ys[p] ? (H - ys[p] + dy[p] - 1) / dy[p] : 0) : H;
        if (!pw || !ph) continue;
        const size_t stride = (pw * bits_pp + 7) / 8;
        if (!unfilter(raw.data() + off, ph, stride, bpp)) return false;
        for (size_t y = 0; y < ph; y++) {
            const uint8_t* line = raw.data() + off + y * (stride + 1) + 1;
            for (size_t x = 0; x < pw; x++) {
                uint8_t r, g, b;
                auto sample = [&](size_t idx) -> unsigned {      // idx-th sample of the line, scaled to 8 bits like stb
                    if (depth == 8) return line[idx];
                    if (depth == 16) return line[2 * idx];       // stb keeps the high byte
                    const size_t bit = idx * depth; const unsigned v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                    return ctype == 3 ? v : v * 255u / ((1u << depth) - 1);
                };
                if (ctype == 0 || ctype == 4) { r = g = b = (uint8_t)sample(x * chans); }
                else if (ctype == 3) {
                    const unsigned i = sample(x);
                    if ((size_t)i * 3 + 2 < plte.size()) { r = plte[3 * i]; g = plte[3 * i + 1]; b = plte[3 * i + 2]; } else { r = g = b = 0; }
                } else { r = (uint8_t)sample(x * chans); g = (uint8_t)sample(x * chans + 1); b = (uint8_t)sample(x * chans + 2); }
                const size_t X = interlace ? xs[p] + x * dx[p] : x, Y = interlace ? ys[p] + y * dy[p] : y;
                uint8_t* o = &rgb[(Y * W + X) * 3];
                o[0] = r; o[1] = g; o[2] = b;
            }
        }
        off += ph * (stride + 1);
    }
    w = (int)W; h = (int)H;
    return true;
}

static bool load_rgb8_unchecked(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h);
bool load_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h) {
    // nothing may throw across the C entry points or out of the CLI's "Failed to load" path (bad_alloc / length_error
    // from a hostile header)
    try { return load_rgb8_unchecked(path, rgb, w, h); } catch (const std::exception&) { rgb.clear(); return false; }
}
static bool load_rgb8_unchecked(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + r);
    fclose(f);
    if (data.size() >= 8 && data[0] == 0x89) return png_decode_rgb8(data.data(), data.size(), rgb, w, h);
    // binary PNM (P5 gray / P6 RGB, maxval 255)
    if (data.size() > 2 && data[0] == 'P' && (data[1] == '5' || data[1] == '6')) {
        size_t pos = 2; long v[3]; int got = 0;
        while (got < 3 && pos < data.size()) {
            while (pos < data.size() && (data[pos] == ' ' || data[pos] == '\n' || data[pos] == '\r' || data[pos] == '\t')) pos++;
            if (pos < data.size() && data[pos] == '#') { while (pos < data.size() && data[pos] != '\n') pos++; continue; }
            long x = 0; bool any = false;
            while (pos < data.size() && data[pos] >= '0' && data[pos] <= '9') { x = x * 10 + (data[pos] - '0'); pos++; any = true; if (x > (1l << 30)) return false; }
            if (!any) return false;
            v[got++] = x;
        }
        pos++;
        if (got < 3 || v[2] != 255 || v[0] < 1 || v[1] < 1 || v[0] > (1l << 24) || v[1] > (1l << 24)) return false;
        const int ch = data[1] == '6' ? 3 : 1;
        if (data.size() - pos < (size_t)v[0] * v[1] * ch) return false;
        w = (int)v[0]; h = (int)v[1];
        rgb.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++)
            for (int c = 0; c < 3; c++) rgb[3 * i + c] = data[pos + i * ch + (ch == 3 ? c : 0)];
        return true;
    }
    return false;
}

bool png_encode_rgb8(const uint8_t* rgb, int w, int h, std::vector<uint8_t>& out) { return png_encode_rgb8_opt(rgb, w, h, out, 6, true); }

// level: zlib level (1 fastest .. 9); adaptive: per-line filter choice (five trial filters per line) or the fixed "up" filter
bool png_encode_rgb8_opt(const uint8_t* rgb, int w, int h, std::vector<uint8_t>& out, int level, bool adaptive) {
    if (w < 1 || h < 1) return false;
    const size_t stride = (size_t)w * 3;
    std::vector<uint8_t> raw((stride + 1) * h), cand(stride);
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < h; y++) {      // per-line filter choice by minimum sum of absolute differences
        const uint8_t* cur = rgb + (size_t)y * stride;
        const uint8_t* prev = y ? rgb + (size_t)(y - 1) * stride : zero.data();
        long best = -1; int bf = 0;
        for (int ft = 0; ft < 5 && adaptive; ft++) {
            long score = 0;
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= 3 ? cur[i - 3] : 0, b = prev[i], c = i >= 3 ? prev[i - 3] : 0;
                const int pr = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) >> 1 : paeth(a, b, c);
                score += abs((int)(int8_t)(uint8_t)(cur[i] - pr));
            }
            if (best < 0 || score < best) { best = score; bf = ft; }
        }
        if (!adaptive) bf = 2;
        uint8_t* o = &raw[(size_t)y * (stride + 1)];
        o[0] = (uint8_t)bf;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= 3 ? cur[i - 3] : 0, b = prev[i], c = i >= 3 ? prev[i - 3] : 0;
            const int pr = bf == 0 ? 0 : bf == 1 ? a : bf == 2 ? b : bf == 3 ? (a + b) >> 1 : paeth(a, b, c);
            o[1 + i] = (uint8_t)(cur[i] - pr);
        }
    }
    uLongf cl = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(cl);
    if (compress2(comp.data(), &cl, raw.data(), (uLong)raw.size(), level < 1 ? 1 : level > 9 ? 9 : level) != Z_OK) return false;
    comp.resize(cl);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.assign(sig, sig + 8);
    auto chunk = [&](const char* type, const std::vector<uint8_t>& body) {
        put_be32(out, (uint32_t)body.size());
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), body.begin(), body.end());
        put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr); chunk("IDAT", comp); chunk("IEND", {});
    return true;
}

bool png_write_rgb8(const std::string& path, const uint8_t* rgb, int w, int h) {
    std::vector<uint8_t> out;
    if (!png_encode_rgb8(rgb, w, h, out)) return false;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

}  // namespace tfh
