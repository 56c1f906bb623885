// tf_png.h -- minimal PNG (and binary PNM) image I/O over zlib for the turtlefft CLI.
// Replaces the reference's use of the vendored stb_image / stb_image_write (S:909, S:1104):
// load any 8/16-bit gray / gray+alpha / RGB / RGBA / palette PNG (interlaced or not) forced to
// 3 channels, write 8-bit RGB.  Other container formats stb can read are out of scope.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tfh {
bool load_rgb8(const std::string& path, std::vector<uint8_t>& rgb, int& w, int& h);
bool png_decode_rgb8(const uint8_t* data, size_t n, std::vector<uint8_t>& rgb, int& w, int& h);
bool png_write_rgb8(const std::string& path, const uint8_t* rgb, int w, int h);
bool png_encode_rgb8(const uint8_t* rgb, int w, int h, std::vector<uint8_t>& out);
bool png_encode_rgb8_opt(const uint8_t* rgb, int w, int h, std::vector<uint8_t>& out, int level, bool adaptive_filter);
}  // namespace tfh
