// tfft_kernels.h -- parameter blocks and launchers shared by tfft_kernels.hip and tfft_capi.hip
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "../../include/turtlefft_hip.h"

namespace tfft {

struct RowParams {
    int W, H;        // image size (pixels)
    int PW, PH;      // padded size (internal PW >= 2)
    int center;      // (-1)^(x+y) pre/post multiply (apply_center)
    float scale;     // inverse only: 1/((PW/2)*PH)
    size_t img_stride;   // float2 elements between consecutive images of a batch in tmp/spec
    float bias;          // forward only: subtracted from every pixel before the transform (its rank-1 transform is added
                         // back by the last column step, ColParams::dc_*); 0 = off
    const uint8_t* cover;    // inverse only, delta embedding: the transform holds IFFT(F' - F) and the pixel is
                             // clamp(round(cover + delta)) (images W*H*3 bytes apart, like the output); nullptr = off, bias is added
};

// a bin of the list, located inside its column tile: value at LDS row `k` (= y / G), column `c` (= x % 16)
struct TileBin { uint16_t k; uint8_t c; uint8_t conj; uint32_t bit; };

struct ColParams {
    int M;             // columns of the half spectrum (PW/2)
    int PH;            // full column length (twiddle table size)
    int G;             // groups (1 for the direct pass, N2 or N1 for the two-step passes)
    int in_a, in_b;    // input row  = in_a*l  + in_b*g
    int out_a, out_b;  // output row = out_a*k + out_b*g
    int in_rows;       // input rows >= in_rows are zero (not loaded)
    int out_rows;      // output rows >= out_rows are not stored
    // extraction straight out of the tiles (k_fft_cols<..., COLS_READ>): the spectrum is never stored
    const struct TileBin* rd_bins;   // bins bucketed by (plane, 16-column tile, group g), see k_bucket_*
    const unsigned* rd_off;          // bucket b = (plane*G + g)*ntiles + tile holds rd_bins[rd_off[b] .. rd_off[b+1])
    uint8_t* rd_bits;                // bits_out, image i at rd_bits + i*rd_n
    uint64_t rd_n;
    uint8_t* trash;                  // >= 8 KiB of device scratch nobody reads: where lanes WITHOUT a list entry send their (unpredicated) stores
    // delta embedding: the LAST forward step (COLS_EMIT) writes the values of the bucketed bins (rd_bins / rd_off as above) to em_fl
    // in bucket order; the FIRST inverse step (COLS_EMBED) starts every tile as zeros and puts F' - F at those bins, F taken from
    // em_fl (S:712-732 with a fixed alpha)
    float2* em_fl;                   // em_n values per image, indexed like rd_bins
    int em_m2;                       // COLS_EMIT: store |F|^2 (float, `out` reinterpreted, same byte offset per image) instead of the complex
                                     // spectrum, the packed column 0 to st_col0: all the statistics read (launch_medians col0_m2)
    const uint8_t* em_pb;            // COLS_EMBED: em_n stream bits per image in the same order (k_gather_bits; 2 = not written)
    uint64_t em_n;                   // list stride between images (the length of the bin list)
    float em_cos, em_sin;
    // statistics inside the last forward step (COLS_STAT = COLS_EMIT without the spectrum store): every value is classified against
    // the bracket of its plane's SelectState exactly as k_collect_bracket does (weight below, candidates, capacity counts, parked
    // values); the packed column 0 goes to st_col0 for k_col0_stats
    struct SelectState* st_sel;      // 3 per image
    unsigned* st_cand; size_t st_cand_stride;
    unsigned st_resv;           // COLS_STAT: slots of the candidate list each wave of the launch owns (set by the launcher: 64 per tile it walks)
    unsigned st_cand_fixed;     // COLS_STAT: slots at the head of a plane's list owned by the launch's waves (set by the launcher); appended entries follow
    unsigned* st_partial;            // per (image, plane) TFFT_STAT_MAX_BLOCKS counters (a workgroup adds to slot block % that)
    float* st_amb;
    float2* st_col0;                 // per (image, plane) PH values
    unsigned st_slo, st_shi;         // squared radius bounds of the annulus, clamped to 32 bits
    int st_cap, st_PW;
    // forward COLS_PLAIN only: a sample of the tiles (0, tile_step, 2*tile_step, ..) written side by side into a narrow spectrum
    int tile_step, tile_off; int out_M; size_t out_plane_stride, out_img_stride;      // tiles tile_off + i*tile_step
    // forward COLS_PLAIN only: images whose statistics were settled without the spectrum (all three planes) return at once
    const struct SelectState* gate;
    // forward COLS_PLAIN sample pass: instead of storing the narrow spectrum, every value's |F|^2 goes into a 4096-bucket histogram (the
    // top 13 bits of the float, weight 2) kept in LDS at byte offset hist_lds_off and added to hist_sel[3*img + plane].hist at the end
    struct SelectState* hist_sel; unsigned hist_lds_off;
    int g_step, g_off;          // ... and only the row groups g_off + i*g_step of the launch (0: all): rows g + G*k, a regular subsample of the rows
    int em_on;
    // DC removal (forward, final step only): out[row][col] += dc_ah[row] * dc_aw[col] -- the transform of the constant that
    // the row kernels subtracted from the pixels, c*A_H(y)*A_W(x); nullptr = off
    const float2* dc_ah;       // PH entries, the factor c included
    const float2* dc_aw;       // M entries, entry 0 packed: A_W(0) + i*A_W(M)
    const int* last_row_dev;   // optional device scalar: rows > *last_row_dev are not stored either (extraction reads
                               // only the rows its bin list touches; k_bins_last_row)
    int tw_out;        // multiply output by exp(sign*2*pi*i*k*g/PH)
    int tiles_per_block;  // adjacent 16-column tiles walked by one workgroup
    size_t plane_stride;  // float2 elements between planes (PH*M)
    size_t img_stride;    // float2 elements between images (grid.z = 3*n_images)
};

struct EmbedParams {
    uint64_t n;
    int PH, PW;
    int generic;       // 0: alpha fixed, no jitter, 0<alpha<pi -> cos/sin constants and sign test
    int adaptive;
    float cos_a, sin_a;
    double alpha;
    double med[3];
    size_t img_stride;     // float2 elements between images (grid.y = image)
    const uint32_t* bit_index;   // bins[i] carries stream bit bit_index[i] (nullptr: bit i); tfft_set_bit_index
    uint64_t limit;              // embed only: stream bits >= limit are not written (the stream is shorter than the bin list)
    // embed only, stream pipelines: the bits come straight out of the packed frame (38-byte header, frame_plen payload bytes per image)
    const uint8_t* frame_hdr; const uint8_t* frame_pay; uint64_t frame_plen;
};

struct CapParams {
    int PH, PW;                     // padded grid as the reference sees it
    int PWi;                        // internal (even) row length used for indexing the half spectrum
    int bw, bh;                     // bounding box of the annulus (x < bw, y < bh)
    unsigned long long s_lo, s_hi;  // s_lo <= y*y+x*x <= s_hi  <=>  rmin*mn <= hypot(y,x) <= rmax*mn
    double magmin;                  // used with med_dev (batch path)
    double thr[3];                  // used when med_dev == nullptr (tfft_capacity)
    size_t img_stride;
};

// Smallest float m2 with (double)sqrtf(m2) >= thr, i.e. the reference's test !(|F| < thr) (S:1004) moved
// onto the argument of mag_of's square root: sqrtf is correctly rounded and monotone, so
// !((double)sqrtf(m2) < thr)  <=>  !(m2 < T2).  One scalar search per thread (wave uniform) replaces a
// correctly rounded square root and a double compare per bin.
__host__ __device__ inline float mag2_threshold(double thr) {
    if (!(thr > 0.0)) return -INFINITY;                 // thr <= 0 or NaN: nothing is ever "< thr"
    float tf = (float)thr;                              // tf = smallest float >= thr
    if ((double)tf < thr) tf = nextafterf(tf, INFINITY);
    if (!(tf < INFINITY)) return INFINITY;
    float c = tf * tf;
    if (!(c < INFINITY)) c = FLT_MAX;
    for (int i = 0; i < 8 && !(sqrtf(c) < tf); i++) c = nextafterf(c, -INFINITY);   // now sqrtf(c) < tf (or c ran to 0)
    for (int i = 0; i < 16 && sqrtf(c) < tf; i++) c = nextafterf(c, INFINITY);      // first value that passes
    return c;
}

// ---- exact medians / capacity of the single-image calls (tfft_exact.hip)
struct ExactCand { uint16_t y, x, w, plane; float m2; };      // a full-grid bin (x = PW/2: the Nyquist column out of the packed column 0), the
                                                              // weight it carries (itself and / or its Hermitian mirror) and its fp32 |F|^2
struct ExactCollect {
    int PH, PW;                       // padded grid (PW = the internal even row length)
    int PW_full;                      // the reference's PW (mirror columns are PW_full - x)
    int cap;                          // 0: median mode, 1: capacity mode (annulus s_lo <= y*y + x*x <= s_hi, axes excluded)
    unsigned long long s_lo, s_hi;
    float lo2[3], hi2[3];             // per plane: window of fp32 |F|^2 whose bins are re-evaluated in fp64
    int cap_cand;                     // candidate slots per plane
};
hipError_t launch_exact_collect(const float2* spec, const ExactCollect& P, ExactCand* cand, unsigned long long* below, unsigned* n_cand, hipStream_t s);
hipError_t launch_exact_table(double2* table, int PW, hipStream_t s);
hipError_t launch_exact_eval(const uint8_t* rgb, int W, int H, int PW, int PH, int center, const ExactCand* cand, unsigned n, int n_split,
                             const double2* table, double2* out, hipStream_t s);

#define TFFT_STAT_MAX_BLOCKS 512

struct SelectState {        // one per (image, plane)
    unsigned hist[4096];
    unsigned long long rank;
    unsigned long long below;   // fast path: exact weight of everything below the bracket
    unsigned prefix;
    unsigned n_cand;
    unsigned lo, hi;            // fast path: bracket of level-1 buckets around the sample median
    unsigned done;              // 0 open / fallback needed, 2 fast path verified at level 2, 1 median written
    unsigned fast;              // 1: the median came from the fast path (set with done = 1 by k_select_fast<3>)
    // capacity counted inside the bracket pass (batch path): the threshold T2 = mag2_threshold(magmin * median) is only
    // known afterwards, but the bracket pins it to [t2_lo, t2_hi]: bins at or above t2_hi count now, bins below t2_lo
    // never, the few in between are parked (their |F|^2, once per full-grid bin) and settled by k_capacity_settle
    float t2_lo, t2_hi;
    unsigned n_amb;             // parked values (may exceed TFFT_AMB_CAP: then the plane falls back to k_capacity)
    unsigned cand_fixed;        // slots at the head of the candidate list owned by COLS_STAT's waves (holes included); appended entries follow
};
#define TFFT_AMB_CAP 8192
#define TFFT_CAND_HOLE 0x7FFFFFFFu      // an unused slot of the candidate list (COLS_STAT reserves slots per wave): weight bit clear, a value no bracket reaches

hipError_t launch_rows_fwd(const uint8_t* rgb, float2* out, const float2* tw_pw, const RowParams& P, int n_images,
                           hipStream_t s);
hipError_t launch_rowcol_fwd(const uint8_t* rgb, float2* out, const float2* tw_pw, const float2* tw_ph, const RowParams& P,
                             int n_images, hipStream_t s);
hipError_t launch_colrow_inv(const float2* in, uint8_t* rgb, const float2* tw_pw, const RowParams& P, int n_images, hipStream_t s);
// the forward stage with threads and LDS for the LIVE rows of every group only (one launch per distinct live-row count)
hipError_t launch_rowcol_fwd_live(const uint8_t* rgb, float2* out, const float2* tw_pw, const float2* tw_ph, const RowParams& P, int n_images, hipStream_t s);
hipError_t launch_rows_inv(const float2* in, uint8_t* rgb, const float2* tw_pw, const RowParams& P, int n_images,
                           hipStream_t s);
hipError_t launch_cols(const float2* in, float2* out, const float2* tw_ph, const ColParams& P, int logl, int sign,
                       int n_planes, hipStream_t s);
// (f-4) fp64 audit transform, tfft_audit64.hip
hipError_t audit_fft2d_f64(double2* a, double2* scratch, double2* wtab, int n_planes, int PH, int PW, int inverse, hipStream_t s);
hipError_t audit_load_rgb8_f64(const uint8_t* rgb_dev, int W, int H, int PW, int PH, int center, double2* out, hipStream_t s);
// bucket the bin list by (plane, 16-column tile, row group y % G) for the tile-resident read: counts -> offsets -> entries
hipError_t launch_bucket_bins(const tfft_bin* bins, const uint32_t* bit_index, uint64_t n, int PH, int PW, int G,
                              unsigned* cnt, unsigned* off, TileBin* out, int* err, int force_global, hipStream_t s);
// highest stored row any bin of the list touches -> *last_row (device int, reset here)
hipError_t launch_stat_guess(const float2* mini, int PH, int PW, int Ms, size_t mini_img_stride, int n_images, SelectState* st, const struct CapParams* cap,
                             unsigned* partial, int col0_packed, hipStream_t s);
hipError_t launch_skew_bracket(SelectState* st, int n_images, int skew, hipStream_t s);
hipError_t launch_stat_select(int PH, int n_images, SelectState* st, unsigned* cand, size_t cand_stride, float* med_out, const float2* col0, hipStream_t s);
hipError_t launch_stat_settle(const float2* spec, int PH, int PW, size_t img_stride, int n_images, SelectState* st, float* med_out, const struct CapParams* cap,
                              unsigned* partial, float* amb, unsigned long long* usable, hipStream_t s);
hipError_t launch_gather_bits(const TileBin* ent, const unsigned* n_ent, const uint8_t* bits, const uint8_t* hdr, const uint8_t* pay, uint64_t plen,
                              uint64_t n, uint64_t limit, int n_images, uint8_t* out, hipStream_t s);
hipError_t launch_bins_last_row(const tfft_bin* bins, uint64_t n, int PH, int PW, int* last_row, hipStream_t s);
hipError_t launch_embed(float2* spec, const tfft_bin* bins, const uint8_t* bits, const float* jitter,
                        const EmbedParams& P, int n_images, int* err, hipStream_t s);
hipError_t launch_read(const float2* spec, const tfft_bin* bins, const float* jitter, const EmbedParams& P,
                       int n_images, uint8_t* bits_out, int* err, hipStream_t s);
// cap != nullptr: also S:998-1008 for every image (magmin in cap->magmin), counted inside the full median pass:
// partial = [n_images*3*TFFT_STAT_MAX_BLOCKS] block counts, amb = [n_images*3*TFFT_AMB_CAP] parked |F|^2, usable[n_images]
hipError_t launch_medians(const float2* spec, int PH, int PW, size_t img_stride, int n_images, SelectState* st,
                          unsigned* cand, size_t cand_stride, float* med_out, int force_fallback, int fill_cus, int fill_resident, hipStream_t s,
                          const CapParams* cap = nullptr, unsigned* partial = nullptr, float* amb = nullptr, unsigned long long* usable = nullptr,
                          int compact = 1, const float2* col0_m2 = nullptr, int skew = 0);
int collect_bracket_resident_blocks();
hipError_t launch_capacity(const float2* spec, const CapParams& P, int n_images, const float* med_dev,
                           unsigned* partial, unsigned long long* usable, hipStream_t s, const unsigned* only_flagged = nullptr);
hipError_t launch_frame_expand(const uint8_t* header, const uint8_t* payload, uint64_t plen, int n_images, uint8_t* bits,
                               uint64_t stride, hipStream_t s);      // image i's bits at bits + i*stride
hipError_t launch_frame_majority(const uint8_t* bits, uint64_t plen, int n_images, uint8_t* header, uint8_t* payload,
                                 hipStream_t s);
hipError_t launch_stream_decode(const uint8_t* bits, uint64_t n_bins, uint64_t max_plen, int n_images, uint8_t* header, uint8_t* payload,
                                int* status, unsigned* plen, hipStream_t s);
hipError_t launch_export_full(const float2* spec, int PH, int PW, int PWout, float2* out, hipStream_t s);
// compute_cover_hash's low-frequency magnitudes in fp64 from the pixels (rowsum: H*3*region double2 of scratch)
hipError_t launch_lowfreq_f64(const uint8_t* rgb, int W, int H, int PW, int PH, int center, int region, double2* rowsum, double* out,
                              hipStream_t s);

}  // namespace tfft
