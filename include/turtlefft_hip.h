/* turtlefft_hip.h -- C ABI of libturtlefft_hip.so, the MI355X (gfx950) replacement
 * for the 2-D-FFT phase-embedding hot path of rickenator/steganosaurus.
 *
 * The reference has no FFI layer: the path is reachable only through the
 * file-static functions of steganosaurus/src/steganosaur.cpp (cited S:<line>).
 * Each entry point below names the reference lines it replaces; INTEGRATION.md
 * shows the patch a maintainer of the reference would apply to do_embed /
 * do_extract to bind them.
 *
 * Conventions
 *   - every function returns TFFT_OK (0) or a negative tfft_status; nothing
 *     calls exit() and nothing throws across the boundary;
 *   - a context owns all device memory, its HIP stream(s) and twiddle tables;
 *     one context is used from one host thread at a time, distinct contexts
 *     (also on distinct devices) may be used concurrently;
 *   - "slot" = one resident image: its three half-spectra stay in HBM between
 *     calls (forward -> medians/capacity -> embed/read -> inverse), exactly the
 *     lifetime of FR/FG/FB/F3 in the reference (S:917-1100);
 *   - pointers are HOST pointers unless the function name ends in _dev, in
 *     which case they are device pointers valid on the context's device;
 *   - transforms use the reference's sign convention: forward kernel
 *     exp(+2*pi*i*nk/N) (S:347), inverse exp(-...) scaled 1/N per dimension
 *     (S:357); images are zero-padded to next_pow2 in each dimension (S:393-398)
 *     and the inverse crops back to W x H (S:399-403);
 *   - there is no CPU fallback: without a usable gfx950 device tfft_create
 *     fails with TFFT_E_NO_DEVICE.
 */
#ifndef TURTLEFFT_HIP_H
#define TURTLEFFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFFT_ABI_VERSION 1

typedef enum tfft_status {
    TFFT_OK = 0,
    TFFT_E_INVALID = -1,      /* bad argument (null pointer, slot out of range, size 0 ...) */
    TFFT_E_NO_DEVICE = -2,    /* no HIP device / not gfx950 / HIP runtime error at create */
    TFFT_E_TOO_LARGE = -3,    /* image larger than the context was created for, or > TFFT_MAX_DIM */
    TFFT_E_NOMEM = -4,        /* host or device allocation failed */
    TFFT_E_HIP = -5,          /* HIP runtime error during a launch/copy (see tfft_last_hip_error) */
    TFFT_E_STATE = -6,        /* slot holds no forward spectrum yet */
    TFFT_E_EXHAUSTED = -7,    /* walk: annulus exhausted (the reference would spin forever) */
    TFFT_E_BIN_RANGE = -8     /* a bin lies outside the padded grid or on an excluded axis */
} tfft_status;

#define TFFT_MAX_DIM 16384     /* largest padded width/height handled (rows of PW/2 complex fit in LDS) */

typedef struct tfft_ctx tfft_ctx;
typedef struct tfft_walk tfft_walk;

/* One embedding position, as produced by Turtle::advance_to_valid (S:778-804):
 * colour plane 0..2 and UN-shifted frequency indices on the padded grid. */
typedef struct tfft_bin {
    uint16_t x;        /* 0 .. PW-1 */
    uint16_t y;        /* 0 .. PH-1 */
    uint8_t plane;     /* 0=R 1=G 2=B */
    uint8_t rsv[3];    /* must be 0 */
} tfft_bin;

/* ------------------------------------------------------------------ context */
int tfft_abi_version(void);
const char* tfft_strerror(int status);

/* Allocates `n_slots` resident images of up to max_w x max_h pixels on HIP
 * device `device`.  Replaces the vectors of S:912-917 / S:1015. */
int tfft_create(int device, int max_w, int max_h, int n_slots, tfft_ctx** out);
int tfft_destroy(tfft_ctx* ctx);
/* Run all work of this context on an existing HIP stream (e.g. the framework's
 * current stream) instead of the context's own.  stream = hipStream_t. */
int tfft_set_stream(tfft_ctx* ctx, void* hip_stream);
int tfft_sync(tfft_ctx* ctx);
int tfft_last_hip_error(const tfft_ctx* ctx);
size_t tfft_device_bytes(const tfft_ctx* ctx);
/* Which kernels a launch over n_images images of w x h takes on this context (measurement / documentation aid):
 * info[0] = 1 when the column transform is one pass, else it is two steps of lengths 2^info[1] and 2^info[2];
 * info[3] = 1 when the row pass is fused with the adjacent column step (a 2-D transform is then two global passes:
 * always for images that pad to 2048 columns, for 4096 columns in launches of two or more images). */
int tfft_plan_info(const tfft_ctx* ctx, int w, int h, int n_images, int info[4]);

/* ------------------------------------------------------------- forward side
 * to_planes_u8 + apply_center + pad_to_fft + fft2d(forward) x3  (S:912-921,
 * S:1116-1123).  rgb = H*W*3 interleaved bytes.  Returns the padded size. */
int tfft_forward_rgb8(tfft_ctx* ctx, int slot, const uint8_t* rgb, int w, int h, int center, int* pw, int* ph);
int tfft_forward_rgb8_dev(tfft_ctx* ctx, int slot, const void* rgb_dev, int w, int h, int center, int* pw, int* ph);

/* median_abs x3 (S:922, S:404-409): element at sorted index P/2 of |F| over
 * the FULL padded plane (mirror bins counted).  Synchronises.
 * EXACT against the reference's fp64 spectrum (to ~1e-13 relative: two fp64 summation orders): the fp32 spectrum locates the
 * element, the handful of bins within 2e-6 of it are re-evaluated in fp64 from the pixels (tfft_exact.hip), and rank and value
 * are settled on those.  Needs the image of the last tfft_forward_rgb8[_dev] of the slot to be still in place (the _dev form
 * keeps the caller's pointer, like tfft_lowfreq_mag) and PW <= 8192; otherwise -- and with TFFT_EXACT_STATS=0 -- the order
 * statistic of the fp32 magnitudes is returned (2e-6 relative).  tfft_exact_info tells which. */
int tfft_medians(tfft_ctx* ctx, int slot, double med[3]);
/* bins per plane the last tfft_medians / tfft_capacity of this context re-evaluated in fp64 (0: the fp32 answer was returned) */
int tfft_exact_info(const tfft_ctx* ctx, int n_fp64[3]);

/* Diagnostic: how the last tfft_medians / batched median of `slot` was obtained, per plane:
 * 1 = sampled bracket + one verified pass (fast path), 0 = full three-level select (fallback). */
int tfft_median_path(tfft_ctx* ctx, int slot, int fast[3]);

/* count_plane (S:998-1008): sum over planes of floor(c/2), c = bins in the
 * annulus [rmin,rmax]*min(PH,PW), off the axes, |F| >= thr[plane].  Synchronises.
 * EXACT like tfft_medians (same conditions): bins within 1e-3 of thr are settled on their fp64 magnitudes, so with
 * thr = magmin * tfft_medians() the count is the reference's integer.  The batched pipelines' usable_out stays the count on the
 * fp32 planes (observed 0-1 off, bound 2: a bin within fp32 rounding of the threshold), see DESIGN.md section 2. */
int tfft_capacity(tfft_ctx* ctx, int slot, double rmin, double rmax, const double thr[3], uint64_t* usable);

/* compute_cover_hash's magnitudes (S:428-436): |F[y][x]| for y,x < region (<= 8) of
 * each plane, out = 3*region*region doubles in plane, y, x order.  Evaluated in fp64
 * straight from the image the slot's last tfft_forward_rgb8[_dev] call read (for the
 * _dev variant that buffer must still be alive): 192 inner products, no transform,
 * so the reference's quantiser floor(log(1+mag)/2) (S:433) sees its own values to
 * ~1e-13 and the 32-byte cover hash is the reference's.  TFFT_E_STATE after a batch
 * call (no single image belongs to the slot).  Synchronises. */
int tfft_lowfreq_mag(tfft_ctx* ctx, int slot, int region, double* out);

/* ------------------------------------------------------------- embed / read
 * The embed loop body, write_bit_on_bin (S:712-732), over a materialised bin
 * list: F[y][x] = polar(max(1e-12,|F|), (bit ? +a : -a) + jitter[i]) and the
 * Hermitian mirror = conj.  a = alpha, or alpha*clamp(|F|/med[plane],0.5,2)
 * when `adaptive` (S:704-710).  jitter may be NULL (all zero).  Bins of one
 * list must be distinct with distinct mirrors (the walk guarantees it). */
int tfft_embed_bins(tfft_ctx* ctx, int slot, const tfft_bin* bins, const uint8_t* bits, const float* jitter,
                    uint64_t n, double alpha, int adaptive, const double med[3]);
int tfft_embed_bins_dev(tfft_ctx* ctx, int slot, const void* bins_dev, const void* bits_dev, const void* jitter_dev,
                        uint64_t n, double alpha, int adaptive, const double med[3]);

/* read_bit_from_bin (S:734-746) over a bin list: bits_out[i] in {0,1}.  May be
 * called repeatedly on a resident spectrum (912 header bits, then the payload:
 * S:1223-1264).  The host variant synchronises. */
int tfft_read_bins(tfft_ctx* ctx, int slot, const tfft_bin* bins, const float* jitter, uint64_t n, double alpha,
                   int adaptive, const double med[3], uint8_t* bits_out);
int tfft_read_bins_dev(tfft_ctx* ctx, int slot, const void* bins_dev, const void* jitter_dev, uint64_t n,
                       double alpha, int adaptive, const double med[3], void* bits_out_dev);

/* ------------------------------------------------------------- inverse side
 * fft2d(inverse) x3 + ifft_crop + apply_center + from_planes_u8 (S:1100-1103):
 * real part, round half away from zero, clamp to 0..255, interleave.  The
 * slot's spectrum is consumed.  The host variant synchronises. */
int tfft_inverse_rgb8(tfft_ctx* ctx, int slot, uint8_t* rgb_out);
int tfft_inverse_rgb8_dev(tfft_ctx* ctx, int slot, void* rgb_out_dev);

/* Parity/debug export: the FULL padded spectra, 3*PH*PW interleaved (re,im)
 * floats, mirror half reconstructed by Hermitian symmetry.  Synchronises. */
int tfft_download_spectrum(tfft_ctx* ctx, int slot, float* out);

/* ---------------------------------------------------- fp64 audit transform
 *   The reference's fft2d (S:341-366) evaluated in double on the
 * device, operation by operation: bit reversal, radix-2 DIT stages, stage
 * twiddles by the recurrence w *= wlen, no FMA contraction, rows then columns,
 * inverse divided by n per dimension.  Its output equals the CPU reference bit
 * for bit (checked against the oracle); it exists to measure the fp32 product
 * path against the reference's arithmetic at sizes where the CPU takes minutes.
 * Slow by design (one global pass per stage).  Host buffers; synchronises.
 *   tfft_audit_fft2d_f64        in place on n_planes planes of ph x pw interleaved
 *                               (re,im) doubles; ph, pw powers of two.
 *   tfft_audit_forward_rgb8_f64 to_planes_u8 + apply_center + pad_to_fft + fft2d
 *                               forward x3 (S:912-921): out = 3*PH*PW*2 doubles. */
int tfft_audit_fft2d_f64(tfft_ctx* ctx, double* planes, int n_planes, int ph, int pw, int inverse);
int tfft_audit_forward_rgb8_f64(tfft_ctx* ctx, const uint8_t* rgb, int w, int h, int center, double* out);

/* ------------------------------------------------------------------ batches
 * n independent images of identical size sharing ONE bin list (the walk does
 * not depend on image content: S:797-799).  Images are processed in chunks of
 * n_slots; every pipeline stage of a chunk is ONE kernel launch over all its
 * images, on the context's stream.  All pointers are device pointers;
 * image i is at rgb + i*w*h*3, its bits at bits + i*n_bits.
 *   embed  : forward -> [medians+capacity when usable_out != NULL] -> embed -> inverse
 *   extract: forward -> read
 * usable_out (device, n uint64) receives each image's capacity so the caller
 * can raise "Message too large" (S:1009-1012) without a sync per image.
 * The embed pipeline uses the linearity of S:1099-1103: stego = clamp(round(cover +
 * IFFT(F' - F))), F' - F being zero but at the bins of the list -- the modified spectrum
 * is never written, the cover buffer is read once more by the last kernel (in-place
 * embedding, rgb_out_dev == rgb_dev, is allowed).  Same image as inverting F' in exact
 * arithmetic; in fp32 it is 1 LSB away from what tfft_embed_bins + tfft_inverse_rgb8
 * return on a few pixels per million (both within 1 LSB of the fp64 reference). */
int tfft_embed_batch_dev(tfft_ctx* ctx, int n_images, const void* rgb_dev, int w, int h, int center,
                         const void* bins_dev, const void* bits_dev, uint64_t n_bits, double alpha,
                         double rmin, double rmax, double magmin, void* usable_out_dev, void* rgb_out_dev);
int tfft_extract_batch_dev(tfft_ctx* ctx, int n_images, const void* rgb_dev, int w, int h, int center,
                           const void* bins_dev, uint64_t n_bits, double alpha, void* bits_out_dev);

/* Stream framing on the device.  expand: for each of n_images, the 38-byte header and
 * payload_len bytes of (ciphertext || tag) become the one-byte-per-bit stream Rep-3(header) || Rep-7(payload),
 * MSB first -- bits_from_bytes + rep3/rep7_encode (S:455-467, S:494-500, S:986-995).  majority: the inverse,
 * rep3/rep7_decode + bytes_from_bits (S:447-454, S:468-474, S:501-508) on 912 + 56*payload_len raw bits per
 * image.  header/payload arrays are packed per image (38 and payload_len bytes apart). */
int tfft_frame_expand_dev(tfft_ctx* ctx, int n_images, const void* header_dev, const void* payload_dev, uint64_t payload_len,
                          void* bits_out_dev);
int tfft_frame_majority_dev(tfft_ctx* ctx, int n_images, const void* bits_dev, uint64_t payload_len, void* header_out_dev,
                            void* payload_out_dev);

/* The same two pipelines on PACKED BYTES, i.e. what do_embed / do_extract do around the walk (f-3 inside the
 * pipelines: only 38 + payload_len bytes per image would have to cross PCIe):
 *   embed  : header (38 B) and payload (ciphertext || tag, payload_len bytes) per image -> Rep-3 / Rep-7 stream
 *            on the device (S:986-995) -> the first 912 + 56*payload_len positions of the caller's walk (n_bins
 *            of them, n_bins >= that, else TFFT_E_INVALID) -> inverse.
 *   extract: forward -> the raw bit of EVERY position of the caller's walk, read in the one pass that has the
 *            spectrum on chip -> Rep-3 majority of the first 912 -> header bytes -> magic, version, clen ->
 *            Rep-7 majority of the next 56*(clen+16) bits of the same walk (S:1223-1264).  The length comes out
 *            of the image, not from the caller.
 *            status_out[i] (int32) = clen >= 0, or -1 "Magic not found.", -2 "Unsupported version" (header byte 4
 *            holds it), -3 the walk (n_bins) or the payload buffer (max_payload_len) is too short for what the
 *            header announces (where the reference keeps walking, S:1260-1264).
 *            header_out: 38 bytes per image (always written; the AAD of S:1299-1301).  payload_out: clen+16 bytes
 *            per image at stride max_payload_len.  raw_bits_out (optional): the n_bins raw bits per image.
 * n_bins is also the length a bit index (tfft_set_bit_index) must have been set for. */
int tfft_embed_stream_batch_dev(tfft_ctx* ctx, int n_images, const void* rgb_dev, int w, int h, int center,
                                const void* bins_dev, uint64_t n_bins, const void* header_dev, const void* payload_dev,
                                uint64_t payload_len, double alpha, double rmin, double rmax, double magmin,
                                void* usable_out_dev, void* rgb_out_dev);
int tfft_extract_stream_batch_dev(tfft_ctx* ctx, int n_images, const void* rgb_dev, int w, int h, int center,
                                  const void* bins_dev, uint64_t n_bins, double alpha, void* header_out_dev,
                                  void* payload_out_dev, uint64_t max_payload_len, void* status_out_dev,
                                  void* raw_bits_out_dev);

/* The same two pipelines for HOST buffers (images packed back to back, one byte per bit): the slots
 * are split into two halves and three HIP streams overlap the PCIe copy-in of the next half-batch,
 * the kernels of the current one and the copy-out of the previous one.  The
 * transfers overlap only when the host buffers are page-locked: tfft_host_alloc / tfft_host_free, or
 * any pinned allocation.  Both calls return after the last result has landed. */
int tfft_embed_batch(tfft_ctx* ctx, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                     const uint8_t* bits, uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                     uint64_t* usable_out /* or NULL */, uint8_t* rgb_out);
int tfft_extract_batch(tfft_ctx* ctx, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                       uint64_t n_bits, double alpha, uint8_t* bits_out);
/* ... and on packed bytes: 38 + payload_len bytes per image go in, 38 + clen + 16 bytes and a status word come out, instead of
 * one byte per stream bit (the framing and the two-phase decode run on the device inside the same three-stream pipeline; semantics
 * as tfft_*_stream_batch_dev; raw_bits_out may be NULL). */
int tfft_embed_stream_batch(tfft_ctx* ctx, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                            uint64_t n_bins, const uint8_t* header, const uint8_t* payload, uint64_t payload_len, double alpha,
                            double rmin, double rmax, double magmin, uint64_t* usable_out /* or NULL */, uint8_t* rgb_out);
int tfft_extract_stream_batch(tfft_ctx* ctx, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                              uint64_t n_bins, double alpha, uint8_t* header_out, uint8_t* payload_out, uint64_t max_payload_len,
                              int32_t* status_out, uint8_t* raw_bits_out /* or NULL */);
void* tfft_host_alloc(size_t bytes);
void tfft_host_free(void* p);

/* --------------------------------------------------------- keyed walk (HOST)
 * KS + Turtle + the density gate (S:665-695, S:749-810, S:1076-1081): a
 * resumable generator of embedding positions.  Pure host code, no device.
 * key_walk = first 32 bytes of HKDF-Expand(path_key, "turtle_keys") (S:1054-1058).
 * tfft_walk_next appends the next n positions; *skipped (optional) is
 * incremented by the density-rejected bins.  Returns TFFT_E_EXHAUSTED instead
 * of spinning when no acceptable bin is left. */
int tfft_walk_create(const uint8_t key_walk[32], int ph, int pw, double rmin, double rmax, double density,
                     tfft_walk** out);
int tfft_walk_next(tfft_walk* w, uint64_t n, tfft_bin* out, uint64_t* skipped);
int tfft_walk_start(const tfft_walk* w, int* plane, int* y, int* x);
uint32_t tfft_walk_ks_blocks(const tfft_walk* w);
int tfft_walk_destroy(tfft_walk* w);
/* KS::jitter (S:690-694) for a materialised list: out[i] = jitter drawn from
 * the plane's own keystream (keys_rgb = 3*32 bytes key_r|key_g|key_b) in list
 * order; two bytes are consumed per bin even when max_jitter == 0 (S:719). */
int tfft_walk_jitter(const uint8_t keys_rgb[96], const tfft_bin* bins, uint64_t n, double max_jitter, float* out);

/* ------------------------------------------------------- bin visiting order
 * (new; no counterpart in the reference, whose loop S:1074-1097 visits the
 * bins in walk order because it materialises them one at a time.)  The walk
 * is a pseudo-random tour, so in walk order every bin is its own DRAM row
 * activation.  The result of embed/read does not depend on the order in which
 * distinct bins are visited (the walk never yields a bin or its mirror twice,
 * S:793-809), so the kernels may visit them in address order:
 *   tfft_bins_sort   (host) sorts `bins` in place by (plane, y, x) and writes
 *                    bit_index[i] = the position bins[i] had in the walk, i.e.
 *                    the stream bit it carries.  Compute jitter (tfft_walk_jitter)
 *                    BEFORE sorting: jitter stays indexed by stream bit.
 *   tfft_set_bit_index  hands that index (host array; copied to the device) to
 *                    the context.  From then on every embed/read/batch call on
 *                    it reads bits[bit_index[i]] / jitter[bit_index[i]] and
 *                    writes bits_out[bit_index[i]] for bins[i]; `bits`,
 *                    `jitter` and `bits_out` keep their stream order, so the
 *                    caller-visible results are identical to the unsorted call.
 *                    The index must be a permutation of 0..n-1 (else
 *                    TFFT_E_INVALID) and calls with a different n fail with
 *                    TFFT_E_STATE until it is cleared with (ctx, NULL, 0). */
int tfft_bins_sort(tfft_bin* bins, uint32_t* bit_index, uint64_t n);
/* Optional promise: the n bins at device pointer bins_dev will not change until they are registered again (or with NULL).  The
 * batch extraction calls that are handed exactly this list then keep what they derive from it -- the per-tile buckets of the
 * tile-resident read, the highest row the list touches -- instead of rebuilding it on every call (5 small launches, ~50 us per
 * 32 x 1080p call).  Results are identical either way; tfft_set_bit_index drops what was kept.  The batched EMBED calls use the
 * same cache (their delta form buckets the list too).  It is keyed on address and length: after rewriting the list in place, register
 * it again.  Bins outside the grid are reported (TFFT_E_BIN_RANGE) by every call on the registered list, not only the first. */
int tfft_bins_register_dev(tfft_ctx* ctx, const void* bins_dev, uint64_t n);
int tfft_set_bit_index(tfft_ctx* ctx, const uint32_t* bit_index, uint64_t n);

/* ------------------------------------------------------------ measurement
 * Device-side timing of whatever was enqueued between the two calls on the
 * context's stream (hipEvent pair on that stream). */
int tfft_timer_begin(tfft_ctx* ctx);
int tfft_timer_end(tfft_ctx* ctx, float* ms);
/* Per-kernel timing for the roofline report: enqueue stage `stage` of the
 * batched pipeline over slots [0, n_images) `reps` times on the context's stream
 * between two HIP events; returns the mean time of ONE repetition (ms) and how
 * many kernel launches one repetition is.  Slot 0 must have been through a
 * forward call (its geometry is reused for the whole batch; results of the
 * repeated stage are discarded by the caller).  Stages: 0 rows_fwd, 1 cols_fwd
 * step A (or the direct column pass), 2 cols_fwd step B, 3 embed, 4 cols_inv
 * step A, 5 cols_inv step B, 6 rows_inv, 7 read, 8 medians (with the capacity
 * count of the batch path inside its full pass), 9 capacity as a pass of its own
 * (0 launches in the default configuration), 10 the final forward column step as
 * extraction runs it. */
int tfft_profile_stage(tfft_ctx* ctx, int n_images, int stage, int reps, const void* rgb_dev, void* rgb_out_dev,
                       const void* bins_dev, const void* bits_dev, void* bits_out_dev, uint64_t n_bits, double alpha,
                       float* ms_per_rep, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* TURTLEFFT_HIP_H */
