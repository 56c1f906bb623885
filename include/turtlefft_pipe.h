/* turtlefft_pipe.h -- PNG files -> GPU -> PNG files for batches of equal-sized covers (libtfpipe.so), the step either side of the
 * hot path: replaces the reference's stbi_load (steganosaur.cpp:909) and stbi_write_png
 * (steganosaur.cpp:1104) around do_embed / do_extract when many images are processed with one key.
 *
 * PNG inflate of chunk k+1, the device work of chunk k (tfft_embed_stream_batch: H2D, kernels, D2H on three HIP streams) and the PNG
 * deflate + file write of chunk k-1 run at the same time: n_threads worker threads do the codec work, the calling thread owns the
 * context.  Crypto stays with the caller as in the stream calls: headers = 38 bytes per image (S:886-904), payloads = ciphertext
 * and tag, payload_len bytes per image.  All files must be w x h (any PNG colour type is forced to RGB8 as stbi_load does).
 *
 * Return: TFFT_OK or a negative tfft_status (TFFT_E_INVALID: a file that is not a w x h image; TFFT_E_STATE: a PNG could not be
 * written).  stage_ms (may be NULL): [0] wall time of the call, [1] summed decode thread time, [2] time inside the device calls,
 * [3] summed encode + write thread time. */
#ifndef TURTLEFFT_PIPE_H
#define TURTLEFFT_PIPE_H
#include "turtlefft_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* png_level: zlib level of the stego PNGs, 1 (fastest, fixed "up" filter) .. 9; 0 or >= 6: per-line adaptive filters as the CLI writes
 * them.  usable_out: n_files capacities (S:998-1008) or NULL. */
int tfp_embed_png_batch(tfft_ctx* ctx, int n_files, const char* const* in_paths, const char* const* out_paths, int w, int h, int center,
                        const tfft_bin* bins, uint64_t n_bins, const uint8_t* headers, const uint8_t* payloads, uint64_t payload_len,
                        double alpha, double rmin, double rmax, double magmin, int chunk_images, int n_threads, int png_level,
                        uint64_t* usable_out, double* stage_ms);
/* headers_out: 38 bytes per file; payloads_out: max_payload_len bytes per file (or NULL); status_out: as tfft_extract_stream_batch */
int tfp_extract_png_batch(tfft_ctx* ctx, int n_files, const char* const* in_paths, int w, int h, int center, const tfft_bin* bins,
                          uint64_t n_bins, double alpha, uint8_t* headers_out, uint8_t* payloads_out, uint64_t max_payload_len,
                          int32_t* status_out, int chunk_images, int n_threads, double* stage_ms);
#ifdef __cplusplus
}
#endif
#endif
