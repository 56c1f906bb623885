// tfft_capi.hip -- context management and the C ABI of libturtlefft_hip.so
// (include/turtlefft_hip.h).  Host C++ only; all device work is in
// tfft_kernels.hip.  There is deliberately no CPU fallback in this file: every
// entry point that computes needs a live gfx950 context.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <complex>
#include <map>
#include <tuple>
#include <new>
#include <vector>

#include "tfft_kernels.h"

extern "C" void tfft_internal_radius_bounds(double lo, double hi, uint64_t* s_lo, uint64_t* s_hi, int* empty);
// host view of the capacity kernel's threshold transform (tests/test_host.py checks it against the reference's compare)
extern "C" float tfft_internal_mag2_threshold(double thr) { return tfft::mag2_threshold(thr); }

namespace {

using namespace tfft;

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }        // S:369
int ilog2i(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

struct Slot {     // geometry of one resident image; its buffers are slices of the context pools
    int W = 0, H = 0, PW = 0, PH = 0, PWi = 0, center = 0;
    bool has_spec = false;
    const uint8_t* rgb_src = nullptr;     // device image the last single-image forward of this slot read (tfft_lowfreq_mag reads it again)
};

struct ColPlan { bool direct; int log_n1, log_n2; bool fused_fwd; };

}  // namespace

struct tfft_ctx {
    int device = 0;
    int max_w = 0, max_h = 0, n_slots = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    int last_hip = 0;
    size_t dev_bytes = 0;
    std::vector<Slot> slots;
    // pools, slot i = slice i (fixed strides so that a run of slots is one batched launch)
    size_t img_stride_b = 0;      // bytes between slots in img_pool
    size_t slot_stride = 0;       // float2 elements between slots in spec_pool / tmp_pool
    size_t cand_stride = 0;       // unsigned elements per (slot, plane) in cand_pool
    uint8_t* img_pool = nullptr;
    float2* spec_pool = nullptr;
    float2* tmp_pool = nullptr;
    unsigned* cand_pool = nullptr;
    int stats_prio = 1;                   // the statistics' side stream at the lowest stream priority
    int stats_tile = 1;                   // batched delta embeds run the statistics' bracket pass inside the last forward column step and never store
                                          // the spectrum or |F|^2 (TFFT_STATS_TILE=0: |F|^2 planes + the statistics kernels over them, round 2's default)
    int stats_tile_step_forced = 0;
    int stats_tile_step = 8;              // every 8th column tile is the sample (TFFT_STATS_TILE_STEP)
    float2* col0_pool = nullptr;          // [n_slots*3*max_ph] the packed column 0 of batched embeds that store |F|^2 planes (ColParams::st_col0)
    int stats_skew = 0;                   // test hook (TFFT_STATS_SKEW): brackets moved by this many buckets -- the fast path fails, the fallbacks run
    SelectState* sel = nullptr;           // [n_slots*3]
    float* med = nullptr;                 // [n_slots*3]
    unsigned* partial = nullptr;          // [n_slots*3*TFFT_STAT_MAX_BLOCKS + n_slots]
    float* amb = nullptr;                 // [n_slots*3*TFFT_AMB_CAP] |F|^2 of the bins the bracket pass could not decide
    unsigned long long* usable = nullptr; // [n_slots]
    // exact medians / capacity of the single-image calls (tfft_exact.hip): candidate lists, counters, fp64 values
    int exact_stats = 1;                  // TFFT_EXACT_STATS=0: tfft_medians / tfft_capacity return the fp32 spectrum's own statistics
    ExactCand* ex_cand = nullptr; double2* ex_val = nullptr; unsigned long long* ex_below = nullptr; unsigned* ex_n = nullptr;
    std::map<int, double2*> ex_table;     // PW -> exp(2 pi i j/PW) in fp64
    int ex_last[3] = {0, 0, 0};           // diagnostics: candidates evaluated per plane by the last exact call (0: fp32 result returned)
    int* err = nullptr;                   // sticky bin-range flag
    uint8_t* trash = nullptr;             // 8 KiB nobody reads: target of the unpredicated list stores of lanes without an entry (ColParams::trash)
    int* last_row = nullptr;              // device scalars of k_bins_last_row, one per compute stream
    // spectrum-free extraction (k_fft_cols<..., COLS_READ>): the bin list bucketed by column tile, per compute stream
    struct TileBuckets { float2* fl = nullptr; uint8_t* pb = nullptr; uint64_t fl_cap = 0;      /* delta embedding: values of the listed bins, n_slots x n (ColParams::em_fl) */
                         unsigned* cnt = nullptr; unsigned* off = nullptr; TileBin* ent = nullptr; uint64_t cap = 0; int nb_cap = 0;
                         // what the buckets / the last-row scalar currently describe (tfft_bins_register_dev: reused while the registered list is the one passed in)
                         const void* built_for = nullptr; uint64_t built_n = 0; int built_ph = 0, built_pw = 0, built_g = 0; const void* built_index = nullptr; bool built_bad = false;
                         const void* row_for = nullptr; uint64_t row_n = 0; int row_ph = 0, row_pw = 0; } tb[2];
    const void* reg_bins = nullptr; uint64_t reg_n = 0;      // tfft_bins_register_dev
    int embed_delta = 1;                  // batched embeds: stego = cover + IFFT(F' - F) (TFFT_EMBED_DELTA=0: write F' into the spectrum and invert it)
    int tile_read = 1;                    // TFFT_TILE_READ=0: row-limited spectrum + k_read always; 1: tile read for chunks of >= 8 images; 3: always; 2: always, with the global-atomic bucket build
    hipStream_t stream2 = nullptr;        // TFFT_STREAMS=2: second half of a batch chunk runs here, concurrently
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // delta embedding leaves the statistics off the critical path (nothing downstream reads the medians): they run on a stream of
    // their own beside the inverse transform -- a read-only pass next to a write-only one (TFFT_STATS_ASYNC=0: in line)
    hipStream_t stream_stats[2] = {nullptr, nullptr};
    hipEvent_t ev_stats_fork[2] = {nullptr, nullptr}, ev_stats_join[2] = {nullptr, nullptr};
    int stats_async = 1;
    int stats_m2 = 1;                     // batched delta embeds store |F|^2 (half the bytes) for the statistics instead of the spectrum nobody else reads (TFFT_STATS_M2=0)
    int n_streams = 1;
    int n_cus = 0, collect_resident = 0;  // grid sizing of the full median pass: fill every CU to the same depth
    uint32_t* bit_index = nullptr;        // tfft_set_bit_index: bins[i] carries stream bit bit_index[i]
    uint64_t bit_index_n = 0;
    std::map<int, float2*> tw;            // N -> table exp(+2 pi i j/N), j < N
    std::map<std::tuple<int, int, int, int>, float2*> dc;   // (valid, N, center, kind) -> DC-removal table, see get_dc_table
    float dc_bias = 128.0f;               // constant taken out of the pixels before the forward transform and put back analytically (see
                                          // get_dc_table; both directions).  ON by default: it is what keeps every coefficient within the
                                          // 1e-4 relative tolerance on padded images.  TFFT_DC_BIAS=0 switches it off (A/B measurements only)
    uint8_t* sio_hdr = nullptr; uint8_t* sio_pay = nullptr; int* sio_status = nullptr; uint64_t sio_plen = 0;      // host-buffer stream pipelines
    uint8_t* stream_bits = nullptr; unsigned* stream_plen = nullptr; size_t stream_cap = 0;   // tfft_*_stream_batch_dev: expanded / raw bits of a chunk
    void* stage_bins = nullptr; void* stage_bits = nullptr; void* stage_jit = nullptr; void* stage_out = nullptr;
    size_t stage_cap = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;      // host-buffer pipeline (created on first use)
    hipEvent_t ev_in[4] = {}, ev_comp[4] = {}, ev_out[4] = {};
    uint8_t* out_pool = nullptr;
    int cols_direct_max_log = 8;          // PH <= 256: one column pass; taller: two-step N1 x N2 (a direct 512 pass reaches 1.8-3.4 TB/s, the two steps 5-6)
    int cols_force_log_n1 = -1;
    int cols_tiles_per_block = 8;
    int cols_tiles_forced = 0;            // TFFT_COLS_TILES given: it also rules the COLS_EMIT step (default there: 2 tiles per workgroup up to L = 256, r3h A/B: 0.601 vs 0.630 ms per 32 x 1080p launch)
    int cols_tiles_embed = 0;             // delta embedding: tiles per workgroup of the first inverse step; 0 = 8 for columns up to 256, 2 from 512 on
                                          // (round 3 A/B, gpurun_out/r3h, per 32 x 1080p launch: 0.490 / 0.437 / 0.404 / 0.394 ms with 2 / 4 / 8 / 16; per 8 x 4K: 0.563 / 0.501 / 0.511 / 0.518 with 1 / 2 / 4 / 8)
    int cols_tiles_stat = 16;             // COLS_STAT: tiles per workgroup (nothing is stored: as the tile-resident read; r4d A/B 1080p x 32: 2.048 / 2.024 / 2.000 / 1.997 ms per embed with 2 / 4 / 8 / 16)
    int cols_tiles_read = 16;             // the tile-resident read walks longer runs (A/B: 0.422 vs 0.455 ms per 32x1080p launch; the storing steps prefer 8)
    int median_force_fallback = 0;
#ifndef TFFT_NO_GRAPHS
    // launch-bound calls (a few images): the launch sequence of a batch call is captured once into a hipGraph, keyed by every
    // argument, and replayed.  state 1 = seen once (the next call captures), exec != nullptr = replay
    struct GraphEntry { hipGraphExec_t exec = nullptr; int state = 0; };
    std::map<std::vector<uint64_t>, GraphEntry> graphs;
#endif
    int stats_fail_once = 0;              // TFFT_STATS_FAIL_ONCE (test hook)
    bool stats_dirty = false;             // a statistics launch sequence broke off midway: the select state (histograms left zero by convention) is cleared before the next one
    bool graphs_stale = false;            // the shared bucket buffers were rebuilt for another list: captured sequences that left the build out must go
    int graph_max_images = 0;             // TFFT_GRAPHS=n: replay calls of up to n images.  Off by default: measured 5 % SLOWER than plain
                                          // launches (0.283 vs 0.267 ms per 1080p round trip) -- a single image is bound by the GPU-side
                                          // latency of its dependent kernels, which a graph does not shorten
    int stats_fused = 1;                  // TFFT_STATS_FUSED=0: capacity as its own pass after the medians (A/B)
    int stats_compact = 1;                // TFFT_STATS_COMPACT=0: the 16-launch statistics pipeline also for small planes (A/B)
    int fuse = 1;
    int fuse_wide = 1;
    int fuse_live = 1;                    // TFFT_FUSE_LIVE=0: the fused forward kernel with a wave (pair) and a slab for all 8 rows of a group (A/B)

    uint8_t* img(int i) const { return img_pool + (size_t)i * img_stride_b; }
    float2* spec(int i) const { return spec_pool + (size_t)i * slot_stride; }
    float2* tmp(int i) const { return tmp_pool + (size_t)i * slot_stride; }
};

namespace {

#define HIPCHK(ctx, call)                                  \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) { (ctx)->last_hip = (int)e_; return TFFT_E_HIP; } \
    } while (0)

int dev_alloc(tfft_ctx* c, void** p, size_t bytes) {
    if (hipMalloc(p, bytes) != hipSuccess) { *p = nullptr; return TFFT_E_NOMEM; }
    c->dev_bytes += bytes;
    return TFFT_OK;
}

int get_twiddles(tfft_ctx* c, int n, const float2** out) {
    auto it = c->tw.find(n);
    if (it != c->tw.end()) { *out = it->second; return TFFT_OK; }
    std::vector<float2> h((size_t)n);
    for (int j = 0; j < n; j++) {
        const double a = 2.0 * M_PI * (double)j / (double)n;
        h[j] = make_float2((float)cos(a), (float)sin(a));
    }
    float2* d = nullptr;
    int rc = dev_alloc(c, (void**)&d, sizeof(float2) * (size_t)n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    c->tw[n] = d;
    *out = d;
    return TFFT_OK;
}

// n = images in the launch the plan is for (the fused 4096-wide kernels only pay off with more than one: 1536 workgroups of
// 1024 threads leave the tail of a single image on a few CUs -- measured 0.85 vs 0.82 ms per 4K round trip)
ColPlan plan_cols(const tfft_ctx* c, int PH, int PWi, int n) {
    const int l = ilog2i(PH);
    ColPlan p;
    p.fused_fwd = false;
    if (c->fuse && c->cols_force_log_n1 < 0 && l >= 7 &&
        ((PWi == 2048 && l - 3 <= 9) || (PWi == 4096 && l - 3 <= 9 && (c->fuse_wide >= 2 || (c->fuse_wide == 1 && n >= 2))))) {
        // rows + first column step in one kernel (k_rowcol_fwd): PH = 8 * N2.  2048 wide: one wave per row; 4096 wide: two waves
        // per row, 1024-thread workgroups, for launches of two or more images (TFFT_FUSE_WIDE=0: never, 2: always)
        p.direct = false; p.log_n1 = 3; p.log_n2 = l - 3; p.fused_fwd = true;
        return p;
    }
    if (l <= c->cols_direct_max_log && c->cols_force_log_n1 < 0) { p.direct = true; p.log_n1 = 0; p.log_n2 = l; return p; }
    p.direct = false;
    int l1 = (c->cols_force_log_n1 >= 0) ? c->cols_force_log_n1 : l / 2;
    if (l1 < 1) l1 = 1;
    if (l1 > l - 1) l1 = l - 1;
    if (l - l1 > 9) l1 = l - 9;
    p.log_n1 = l1; p.log_n2 = l - l1;
    return p;
}

int set_geometry(tfft_ctx* c, Slot& s, int w, int h, int center) {
    if (w < 1 || h < 1) return TFFT_E_INVALID;
    if (w > c->max_w || h > c->max_h) return TFFT_E_TOO_LARGE;
    s.W = w; s.H = h; s.PW = next_pow2(w); s.PH = next_pow2(h);
    s.PWi = s.PW < 2 ? 2 : s.PW;          // the real<->half-complex row transform needs an even length
    s.center = center ? 1 : 0;
    if (s.PWi > TFFT_MAX_DIM || s.PH > TFFT_MAX_DIM) return TFFT_E_TOO_LARGE;
    return TFFT_OK;
}

// The pipeline as addressable stages, each ONE batched launch over slots [s0, s0+n) of equal geometry
// (also used by tfft_profile_stage).
//   forward : ROWS_FWD (u8 -> tmp), COLS_FWD_A (tmp -> tmp | spec), COLS_FWD_B (tmp -> spec, two-step only)
//   inverse : COLS_INV_A (spec -> tmp), COLS_INV_B (tmp -> tmp, two-step only), ROWS_INV (tmp -> u8)
enum Stage { ROWS_FWD = 0, COLS_FWD_A = 1, COLS_FWD_B = 2, EMBED = 3, COLS_INV_A = 4, COLS_INV_B = 5, ROWS_INV = 6,
             READ = 7, MEDIANS = 8, CAPACITY = 9,
             COLS_FWD_READ = 10,      // the final forward column step as extraction runs it (rows above the bin list's last row not stored)
             N_STAGES = 11 };

int get_dc_table(tfft_ctx* c, int valid, int N, int center, int kind, double scale, const float2** out);
void invalidate_graphs(tfft_ctx* c);      // cached launch sequences hold raw device pointers: dropped whenever a buffer is reallocated

static void copy_embed_fields(ColParams& cp, const ColParams& e) {
    cp.rd_bins = e.rd_bins; cp.rd_off = e.rd_off; cp.trash = e.trash;
    cp.em_n = e.em_n; cp.em_cos = e.em_cos; cp.em_sin = e.em_sin; cp.em_fl = e.em_fl; cp.em_pb = e.em_pb; cp.em_on = 1; cp.em_m2 = e.em_m2;
    cp.st_sel = e.st_sel; cp.st_cand = e.st_cand; cp.st_cand_stride = e.st_cand_stride; cp.st_partial = e.st_partial; cp.st_amb = e.st_amb;
    cp.st_col0 = e.st_col0; cp.st_slo = e.st_slo; cp.st_shi = e.st_shi; cp.st_cap = e.st_cap; cp.st_PW = e.st_PW;
}

static void copy_plain_extra(ColParams& cp, const ColParams& e) {
    if (e.tile_step > 1) cp.tiles_per_block = e.hist_sel ? 2 : 1;      // the sample: an eighth of the tiles; few per workgroup keep the grid wide
    cp.tile_step = e.tile_step; cp.tile_off = e.tile_off; cp.out_M = e.out_M; cp.out_plane_stride = e.out_plane_stride; cp.out_img_stride = e.out_img_stride; cp.gate = e.gate;
    cp.hist_sel = e.hist_sel; cp.g_step = e.g_step; cp.g_off = e.g_off;
}

// How a launch sequence wants the outer column steps and the inverse row kernel to run: handed down explicitly per call (until round 3
// these were pointers parked in the context around a call -- to stack objects, and left dangling by any early return in between).
struct StageMode {
    const ColParams* fwd_emit = nullptr;   // the last forward column step also writes the values of the listed bins (COLS_EMIT, delta embedding)
    const ColParams* fwd_read = nullptr;   // the last forward column step runs in COLS_READ mode with these rd_* fields (no spectrum stored)
    const int* fwd_last_row = nullptr;     // the last forward column step stores rows <= *fwd_last_row only (COLS_ROWLIMIT)
    const ColParams* inv_embed = nullptr;  // the first inverse column step runs in COLS_EMBED mode (delta embedding) with these rd_*/em_* fields
    const uint8_t* inv_cover = nullptr;    // ... and the inverse row kernel adds its transform to these cover pixels
    const ColParams* fwd_plain_extra = nullptr;   // plain last forward step: tile_step / out_* (the statistics' sample) or gate fields, and ...
    float2* fwd_out_override = nullptr;           // ... its output buffer
    bool inv_via_spec = false;             // delta embedding: the inverse keeps its intermediate in `spec` (nothing reads F there), so `tmp` -- the
                                           // input of the last forward step -- survives for the gated fallback of the in-kernel statistics
};

int enqueue_fft_stage(tfft_ctx* c, int s0, int n, int stage, const uint8_t* rgb_in, uint8_t* rgb_out, hipStream_t st, const StageMode& md = StageMode()) {
    const Slot& s = c->slots[s0];
    const int M = s.PWi / 2;
    const float2 *tw_w, *tw_h;
    int rc = get_twiddles(c, s.PWi, &tw_w); if (rc) return rc;
    rc = get_twiddles(c, s.PH, &tw_h); if (rc) return rc;
    const ColPlan pl = plan_cols(c, s.PH, s.PWi, n);
    const int N1 = 1 << pl.log_n1, N2 = 1 << pl.log_n2;
    float2 *spec = c->spec(s0), *tmp = c->tmp(s0);
    ColParams cp{};
    cp.M = M; cp.PH = s.PH; cp.plane_stride = (size_t)s.PH * M; cp.img_stride = c->slot_stride;
    cp.tiles_per_block = c->cols_tiles_per_block;
    switch (stage) {
        case ROWS_FWD: {
            RowParams rp{s.W, s.H, s.PWi, s.PH, s.center, 0.f, c->slot_stride, c->dc_bias, nullptr};
            // rows + column step A; when some rows of the groups are padding, by the live-rows-only kernel (a full-height image has none)
            if (pl.fused_fwd && c->fuse_live && s.H < s.PH) HIPCHK(c, launch_rowcol_fwd_live(rgb_in, tmp, tw_w, tw_h, rp, n, st));
            else if (pl.fused_fwd) HIPCHK(c, launch_rowcol_fwd(rgb_in, tmp, tw_w, tw_h, rp, n, st));
            else HIPCHK(c, launch_rows_fwd(rgb_in, tmp, tw_w, rp, n, st));
            return TFFT_OK;
        }
        case COLS_FWD_A:
            if (pl.fused_fwd) return TFFT_OK;       // done inside ROWS_FWD
            if (pl.direct) {
                cp.G = 1; cp.in_a = 1; cp.in_b = 0; cp.out_a = 1; cp.out_b = 0; cp.in_rows = s.H; cp.out_rows = s.PH; cp.tw_out = 0;
                cp.last_row_dev = md.fwd_last_row;
            if (c->dc_bias != 0.0f) {
                rc = get_dc_table(c, s.H, s.PH, s.center, 0, (double)c->dc_bias, &cp.dc_ah); if (rc) return rc;
                rc = get_dc_table(c, s.W, s.PWi, s.center, 1, 1.0, &cp.dc_aw); if (rc) return rc;
            }
                if (md.fwd_read) { const ColParams& r = *md.fwd_read; cp.rd_bins = r.rd_bins; cp.rd_off = r.rd_off; cp.rd_bits = r.rd_bits; cp.rd_n = r.rd_n; cp.trash = c->trash; cp.tiles_per_block = c->cols_tiles_read; }
                else if (md.fwd_emit) { copy_embed_fields(cp, *md.fwd_emit); if (!c->cols_tiles_forced && pl.log_n2 <= 8) cp.tiles_per_block = 2; if (cp.st_sel && c->cols_tiles_stat) cp.tiles_per_block = c->cols_tiles_stat; }
                else if (md.fwd_plain_extra) copy_plain_extra(cp, *md.fwd_plain_extra);
                HIPCHK(c, launch_cols(tmp, md.fwd_out_override ? md.fwd_out_override : spec, tw_h, cp, pl.log_n2, +1, 3 * n, st));
            } else {   // for every n2: length-N1 FFT over rows n1*N2+n2, times w^(n2*k1), in place
                cp.G = N2; cp.in_a = N2; cp.in_b = 1; cp.out_a = N2; cp.out_b = 1; cp.in_rows = s.H; cp.out_rows = s.PH; cp.tw_out = 1;
                HIPCHK(c, launch_cols(tmp, tmp, tw_h, cp, pl.log_n1, +1, 3 * n, st));
            }
            return TFFT_OK;
        case COLS_FWD_B:
            if (pl.direct) return TFFT_OK;
            // for every k1: length-N2 FFT over rows k1*N2+n2 -> rows k1+N1*k2
            cp.G = N1; cp.in_a = 1; cp.in_b = N2; cp.out_a = N1; cp.out_b = 1; cp.in_rows = s.PH; cp.out_rows = s.PH; cp.tw_out = 0;
            cp.last_row_dev = md.fwd_last_row;
            if (c->dc_bias != 0.0f) {
                rc = get_dc_table(c, s.H, s.PH, s.center, 0, (double)c->dc_bias, &cp.dc_ah); if (rc) return rc;
                rc = get_dc_table(c, s.W, s.PWi, s.center, 1, 1.0, &cp.dc_aw); if (rc) return rc;
            }
            if (md.fwd_read) { const ColParams& r = *md.fwd_read; cp.rd_bins = r.rd_bins; cp.rd_off = r.rd_off; cp.rd_bits = r.rd_bits; cp.rd_n = r.rd_n; cp.trash = c->trash; cp.tiles_per_block = c->cols_tiles_read; }
            else if (md.fwd_emit) { copy_embed_fields(cp, *md.fwd_emit); if (!c->cols_tiles_forced && pl.log_n2 <= 8) cp.tiles_per_block = 2; if (cp.st_sel && c->cols_tiles_stat) cp.tiles_per_block = c->cols_tiles_stat; }
                else if (md.fwd_plain_extra) copy_plain_extra(cp, *md.fwd_plain_extra);
            HIPCHK(c, launch_cols(tmp, md.fwd_out_override ? md.fwd_out_override : spec, tw_h, cp, pl.log_n2, +1, 3 * n, st));
            return TFFT_OK;
        case COLS_INV_A:
            if (pl.direct) {
                cp.G = 1; cp.in_a = 1; cp.in_b = 0; cp.out_a = 1; cp.out_b = 0; cp.in_rows = s.PH; cp.out_rows = s.H; cp.tw_out = 0;
                if (md.inv_embed) { copy_embed_fields(cp, *md.inv_embed); cp.tiles_per_block = c->cols_tiles_embed ? c->cols_tiles_embed : (pl.log_n2 >= 9 ? 2 : 16); }
                else if (c->dc_bias != 0.0f) {
                    rc = get_dc_table(c, s.H, s.PH, s.center, 0, (double)c->dc_bias, &cp.dc_ah); if (rc) return rc;
                    rc = get_dc_table(c, s.W, s.PWi, s.center, 1, 1.0, &cp.dc_aw); if (rc) return rc;
                }
                HIPCHK(c, launch_cols(spec, md.inv_via_spec ? spec : tmp, tw_h, cp, pl.log_n2, -1, 3 * n, st));
            } else {   // for every k1: length-N2 inverse over rows k1+N1*k2 -> rows k1*N2+n2, times w^-(n2*k1)
                cp.G = N1; cp.in_a = N1; cp.in_b = 1; cp.out_a = 1; cp.out_b = N2; cp.in_rows = s.PH; cp.out_rows = s.PH; cp.tw_out = 1;
                if (md.inv_embed) { copy_embed_fields(cp, *md.inv_embed); cp.tiles_per_block = c->cols_tiles_embed ? c->cols_tiles_embed : (pl.log_n2 >= 9 ? 2 : 16); }
                else if (c->dc_bias != 0.0f) {
                    rc = get_dc_table(c, s.H, s.PH, s.center, 0, (double)c->dc_bias, &cp.dc_ah); if (rc) return rc;
                    rc = get_dc_table(c, s.W, s.PWi, s.center, 1, 1.0, &cp.dc_aw); if (rc) return rc;
                }
                HIPCHK(c, launch_cols(spec, md.inv_via_spec ? spec : tmp, tw_h, cp, pl.log_n2, -1, 3 * n, st));
            }
            return TFFT_OK;
        case COLS_INV_B:
            if (pl.direct || pl.fused_fwd) return TFFT_OK;      // fused: done inside ROWS_INV
            // for every n2: length-N1 inverse over rows k1*N2+n2 -> rows n1*N2+n2 (< H only), in place
            cp.G = N2; cp.in_a = N2; cp.in_b = 1; cp.out_a = N2; cp.out_b = 1; cp.in_rows = s.PH; cp.out_rows = s.H; cp.tw_out = 0;
            HIPCHK(c, launch_cols(md.inv_via_spec ? spec : tmp, md.inv_via_spec ? spec : tmp, tw_h, cp, pl.log_n1, -1, 3 * n, st));
            return TFFT_OK;
        case ROWS_INV: {
            RowParams rp{s.W, s.H, s.PWi, s.PH, s.center, (float)(1.0 / ((double)M * (double)s.PH)), c->slot_stride, c->dc_bias, nullptr};
            if (md.inv_cover) { rp.cover = md.inv_cover; rp.bias = 0.f; }      // delta embedding: the transform of F' - F has no DC term to give back
            if (md.inv_via_spec && !md.inv_embed) return TFFT_E_INVALID;      // only the delta inverse reads nothing from `spec`
            if (pl.fused_fwd) HIPCHK(c, launch_colrow_inv(md.inv_via_spec ? spec : tmp, rgb_out, tw_w, rp, n, st));       // column step B' + rows
            else HIPCHK(c, launch_rows_inv(md.inv_via_spec ? spec : tmp, rgb_out, tw_w, rp, n, st));
            return TFFT_OK;
        }
        default: return TFFT_E_INVALID;
    }
}

// forward: rows (u8 -> tmp) then columns (tmp -> spec) for slots [s0, s0+n); images contiguous at rgb_dev
int enqueue_forward(tfft_ctx* c, int s0, int n, const uint8_t* rgb_dev, hipStream_t st, const StageMode& md = StageMode()) {
    for (int stage : {ROWS_FWD, COLS_FWD_A, COLS_FWD_B}) {
        int rc = enqueue_fft_stage(c, s0, n, stage, rgb_dev, nullptr, st, md);
        if (rc) return rc;
    }
    for (int i = 0; i < n; i++) { c->slots[s0 + i].has_spec = true; c->slots[s0 + i].rgb_src = nullptr; }
    return TFFT_OK;
}

// inverse: columns (spec -> tmp, only rows < H kept) then rows (tmp -> u8)
int enqueue_inverse(tfft_ctx* c, int s0, int n, uint8_t* rgb_out_dev, hipStream_t st, const StageMode& md = StageMode()) {
    for (int stage : {COLS_INV_A, COLS_INV_B, ROWS_INV}) {
        int rc = enqueue_fft_stage(c, s0, n, stage, nullptr, rgb_out_dev, st, md);
        if (rc) return rc;
    }
    for (int i = 0; i < n; i++) c->slots[s0 + i].has_spec = false;
    return TFFT_OK;
}

// DC removal.  A DC-heavy image (every photograph) makes the partial sums of an fp32 FFT as large as the mean
// term itself, and the rows / columns through the DC bin -- and the bins beside them, which sit on its sidelobes
// when the image is padded -- come out with an ABSOLUTE error of ~1 ulp of that term (the fp64 audit transform
// measured 1e-4..4e-4 of the spectrum's rms on the axes, 1.5e-4 relative beside them).  So the row kernels subtract
// a constant c from every pixel (s*(b - c), s = the centring sign) and the LAST forward column step adds the exact
// transform of s*c*rect(W x H) back:  c * A_H(y) * A_W(x),  A_N(k) = sum_{n < valid} sigma^n exp(+2 pi i n k/N),
// a geometric series evaluated in fp64 here.  kind 0: c*A_H(y), y < PH.  kind 1: A_W(x), x < M, entry 0 packed as
// A_W(0) + i*A_W(M) like the spectrum's column 0.
static std::complex<double> dc_series(int k, int valid, int N, int center) {
    double th = 2.0 * M_PI * (double)k / (double)N + (center ? M_PI : 0.0);
    th = fmod(th, 2.0 * M_PI);
    const double sh = sin(0.5 * th);
    if (fabs(sh) < 1e-14) return std::complex<double>((double)valid, 0.0);
    const double mag = sin(0.5 * th * valid) / sh, ph = 0.5 * th * (valid - 1);
    return std::complex<double>(mag * cos(ph), mag * sin(ph));
}
int get_dc_table(tfft_ctx* c, int valid, int N, int center, int kind, double scale, const float2** out) {
    const auto key = std::make_tuple(valid, N, center ? 1 : 0, kind);
    auto it = c->dc.find(key);
    if (it != c->dc.end()) { *out = it->second; return TFFT_OK; }
    const int n = (kind == 0) ? N : (N / 2 > 0 ? N / 2 : 1);
    std::vector<float2> h((size_t)n);
    for (int k = 0; k < n; k++) {
        std::complex<double> a = dc_series(k, valid, N, center);
        if (kind == 1 && k == 0) a += std::complex<double>(0.0, 1.0) * dc_series(N / 2, valid, N, center);
        a *= scale;
        h[k] = make_float2((float)a.real(), (float)a.imag());
    }
    float2* d = nullptr;
    int rc = dev_alloc(c, (void**)&d, sizeof(float2) * (size_t)n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    c->dc[key] = d;
    *out = d;
    return TFFT_OK;
}

EmbedParams embed_params(const tfft_ctx* c, const Slot& s, uint64_t n, double alpha, int adaptive, const double med[3],
                         bool has_jitter) {
    EmbedParams p{};
    p.n = n; p.PH = s.PH; p.PW = s.PW;
    p.adaptive = adaptive ? 1 : 0;
    p.generic = (adaptive || has_jitter || !(alpha > 0.0 && alpha < M_PI)) ? 1 : 0;
    p.cos_a = (float)cos(alpha); p.sin_a = (float)sin(alpha);
    p.alpha = alpha;
    for (int i = 0; i < 3; i++) p.med[i] = med ? med[i] : 0.0;
    p.img_stride = c->slot_stride;
    p.bit_index = c->bit_index;            // callers check index_ok(c, n) first
    p.limit = n;
    return p;
}
// a bit index, once set, must describe exactly the bin list it is used with
static inline bool index_ok(const tfft_ctx* c, uint64_t n) { return !c->bit_index || c->bit_index_n == n; }

CapParams cap_params(const tfft_ctx* c, const Slot& s, double rmin, double rmax) {
    CapParams p{};
    p.PH = s.PH; p.PW = s.PW; p.PWi = s.PWi; p.img_stride = c->slot_stride;
    const int mn = s.PH < s.PW ? s.PH : s.PW;
    const double lo = rmin * mn, hi = rmax * mn;        // S:1003
    uint64_t a, b; int empty;
    tfft_internal_radius_bounds(lo, hi, &a, &b, &empty);
    p.s_lo = a; p.s_hi = b;
    if (empty) { p.bw = p.bh = 0; return p; }
    double lim = floor(hi) + 1.0;
    p.bw = (int)(lim < (double)s.PW ? lim : (double)s.PW);
    p.bh = (int)(lim < (double)s.PH ? lim : (double)s.PH);
    return p;
}

// medians of slots [s0, s0+n); cap != nullptr: also their capacities (S:998-1008 with thr = magmin * median) -> usable[0..n)
// m2: the slots hold |F|^2 planes + packed columns 0 (stats_m2_applies) instead of the spectrum
// The select kernels leave SelectState.hist zero behind them instead of clearing it in front (one launch less per call): a sequence that
// breaks off midway -- a failed launch, an error on the side stream -- would hand dirty histograms to every later call on those slots.
// Whoever sees such a failure marks the context; the next statistics sequence clears the whole state first.
static int stats_clean_if_dirty(tfft_ctx* c, hipStream_t st) {
    if (!c->stats_dirty) return TFFT_OK;
    HIPCHK(c, hipMemsetAsync(c->sel, 0, (size_t)c->n_slots * 3 * sizeof(SelectState), st));
    c->stats_dirty = false;
    return TFFT_OK;
}
static int enqueue_medians_impl(tfft_ctx* c, int s0, int n, hipStream_t st, const CapParams* cap, unsigned long long* usable, bool m2);
int enqueue_medians(tfft_ctx* c, int s0, int n, hipStream_t st, const CapParams* cap = nullptr, unsigned long long* usable = nullptr, bool m2 = false) {
    int rc = stats_clean_if_dirty(c, st);
    if (!rc) rc = enqueue_medians_impl(c, s0, n, st, cap, usable, m2);
    if (rc) c->stats_dirty = true;
    return rc;
}
static int enqueue_medians_impl(tfft_ctx* c, int s0, int n, hipStream_t st, const CapParams* cap, unsigned long long* usable, bool m2) {
    const Slot& s = c->slots[s0];
    // the batch capacity keeps its partial counts and flags in ONE region per call: slots [s0, s0+n) use the start of the pool's
    // share of the compute stream (s0 is 0 or the second half of a two-stream chunk: shares do not overlap for n <= n_slots - s0)
    unsigned* partial = c->partial + (size_t)s0 * (3 * TFFT_STAT_MAX_BLOCKS + 1);
    HIPCHK(c, launch_medians(c->spec(s0), s.PH, s.PWi, c->slot_stride, n, c->sel + 3 * s0,
                             c->cand_pool + (size_t)3 * s0 * c->cand_stride, c->cand_stride, c->med + 3 * s0, c->median_force_fallback, c->n_cus, c->collect_resident, st,
                             cap, partial, c->amb + (size_t)3 * s0 * TFFT_AMB_CAP, usable, c->stats_compact,
                             m2 ? c->col0_pool + (size_t)s0 * 3 * s.PH : nullptr, c->stats_skew));
    return TFFT_OK;
}
// the batched delta embeds with capacity: may the last forward step store |F|^2 instead of the spectrum?
static bool stats_m2_applies(const tfft_ctx* c, const Slot& s, const CapParams& p) {
    return c->stats_m2 && c->stats_fused && c->stats_compact && !c->median_force_fallback && p.bw > 0 &&
           (unsigned long long)s.PH * s.PWi <= (1ull << 24);
}

int ensure_stage(tfft_ctx* c, uint64_t n) {
    if (n <= c->stage_cap) return TFFT_OK;
    (void)hipStreamSynchronize(c->stream);
    invalidate_graphs(c);
    if (c->stage_bins) { (void)hipFree(c->stage_bins); (void)hipFree(c->stage_bits); (void)hipFree(c->stage_jit); (void)hipFree(c->stage_out); }
    c->stage_bins = c->stage_bits = c->stage_jit = c->stage_out = nullptr; c->stage_cap = 0;
    size_t cap = (size_t)n + (size_t)n / 4 + 1024;
    if (dev_alloc(c, &c->stage_bins, cap * sizeof(tfft_bin)) || dev_alloc(c, &c->stage_bits, cap) ||
        dev_alloc(c, &c->stage_jit, cap * sizeof(float)) || dev_alloc(c, &c->stage_out, cap))
        return TFFT_E_NOMEM;
    c->stage_cap = cap;
    return TFFT_OK;
}

int check_err_flag(tfft_ctx* c) {
    int flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flag) {
        HIPCHK(c, hipMemsetAsync(c->err, 0, sizeof(int), c->stream));
        return TFFT_E_BIN_RANGE;
    }
    return TFFT_OK;
}

// ---- hipGraph replay of launch-bound batch calls ---------------------------------------------------------------
void invalidate_graphs(tfft_ctx* c) {
#ifndef TFFT_NO_GRAPHS
    for (auto& kv : c->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    c->graphs.clear();
#else
    (void)c;
#endif
}
inline uint64_t key_bits(double v) { uint64_t u; memcpy(&u, &v, sizeof u); return u; }
inline uint64_t key_bits(const void* p) { return (uint64_t)(uintptr_t)p; }
// enqueue(): the normal launch sequence on c->stream.  after(): the host-side slot state the sequence leaves behind (replays skip
// the host code of enqueue()).  First call with a key: plain launches (every table / buffer the sequence needs is created here,
// outside any capture).  Second call: captured + instantiated + launched.  Later calls: one hipGraphLaunch.
template <class Enqueue, class After>
int with_graph(tfft_ctx* c, int n_images, const std::vector<uint64_t>& key, Enqueue&& enqueue, After&& after) {
#ifndef TFFT_NO_GRAPHS
    if (c->graph_max_images > 0 && n_images > 0 && n_images <= c->graph_max_images && c->n_streams < 2) {
        if (c->graphs.size() > 64 || c->graphs_stale) { invalidate_graphs(c); c->graphs_stale = false; }
        auto& e = c->graphs[key];
        if (e.exec) {
            HIPCHK(c, hipGraphLaunch(e.exec, c->stream));
            return after();
        }
        if (e.state == 1) {
            e.state = 2;            // whatever happens, do not try to capture this key again
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = enqueue();
                hipGraph_t g = nullptr;
                const hipError_t ee = hipStreamEndCapture(c->stream, &g);
                hipGraphExec_t ex = nullptr;
                if (rc == TFFT_OK && ee == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
                    (void)hipGraphDestroy(g);
                    c->graphs[key].exec = ex;
                    HIPCHK(c, hipGraphLaunch(ex, c->stream));
                    return after();
                }
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                c->last_hip = 0;
                return enqueue();       // capture refused or broken (nothing ran): plain launches; a genuine error shows up again here
            }
            (void)hipGetLastError();
        } else if (e.state == 0) e.state = 1;
    }
#else
    (void)n_images; (void)key; (void)after;
#endif
    return enqueue();
}

bool slot_ok(const tfft_ctx* c, int slot) { return c && slot >= 0 && slot < c->n_slots; }

}  // namespace

extern "C" {

int tfft_abi_version(void) { return TFFT_ABI_VERSION; }

const char* tfft_strerror(int status) {
    switch (status) {
        case TFFT_OK: return "ok";
        case TFFT_E_INVALID: return "invalid argument";
        case TFFT_E_NO_DEVICE: return "no usable gfx950 HIP device (this library has no CPU fallback)";
        case TFFT_E_TOO_LARGE: return "image larger than the context allows";
        case TFFT_E_NOMEM: return "out of memory";
        case TFFT_E_HIP: return "HIP runtime error";
        case TFFT_E_STATE: return "slot holds no forward spectrum";
        case TFFT_E_EXHAUSTED: return "annulus exhausted";
        case TFFT_E_BIN_RANGE: return "bin outside the grid or on an excluded axis";
        default: return "unknown status";
    }
}

int tfft_create(int device, int max_w, int max_h, int n_slots, tfft_ctx** out) {
    if (!out || max_w < 1 || max_h < 1 || n_slots < 1 || n_slots > 1024) return TFFT_E_INVALID;
    *out = nullptr;
    int pw = next_pow2(max_w), ph = next_pow2(max_h);
    if (pw < 2) pw = 2;
    if (pw > TFFT_MAX_DIM || ph > TFFT_MAX_DIM) return TFFT_E_TOO_LARGE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return TFFT_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return TFFT_E_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("TFFT_ALLOW_OTHER_ARCH")) return TFFT_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TFFT_E_NO_DEVICE;
    tfft_ctx* c = new (std::nothrow) tfft_ctx();
    if (!c) return TFFT_E_NOMEM;
    c->device = device; c->max_w = max_w; c->max_h = max_h; c->n_slots = n_slots;
    if (const char* e = getenv("TFFT_COLS_DIRECT_MAX_LOG")) c->cols_direct_max_log = atoi(e);
    if (const char* e = getenv("TFFT_COLS_LOG_N1")) c->cols_force_log_n1 = atoi(e);
    c->n_cus = prop.multiProcessorCount;
    c->collect_resident = collect_bracket_resident_blocks();
    if (const char* e = getenv("TFFT_FUSE")) c->fuse = atoi(e);
    if (const char* e = getenv("TFFT_FUSE_WIDE")) c->fuse_wide = atoi(e);
    if (const char* e = getenv("TFFT_FUSE_LIVE")) c->fuse_live = atoi(e);
    if (const char* e = getenv("TFFT_EMBED_DELTA")) c->embed_delta = atoi(e);
    if (const char* e = getenv("TFFT_STATS_ASYNC")) c->stats_async = atoi(e);
    if (const char* e = getenv("TFFT_STATS_M2")) c->stats_m2 = atoi(e);
    if (const char* e = getenv("TFFT_STATS_SKEW")) c->stats_skew = atoi(e);
    if (const char* e = getenv("TFFT_STREAMS")) c->n_streams = atoi(e);
    if (const char* e = getenv("TFFT_TILE_READ")) c->tile_read = atoi(e);
    if (const char* e = getenv("TFFT_DC_BIAS")) c->dc_bias = (float)atof(e);
    if (const char* e = getenv("TFFT_MEDIAN_FALLBACK")) c->median_force_fallback = atoi(e);
    if (const char* e = getenv("TFFT_STATS_FUSED")) c->stats_fused = atoi(e);
    if (const char* e = getenv("TFFT_STATS_COMPACT")) c->stats_compact = atoi(e);
    if (const char* e = getenv("TFFT_GRAPHS")) c->graph_max_images = atoi(e);
    if (const char* e = getenv("TFFT_EXACT_STATS")) c->exact_stats = atoi(e);
    if (const char* e = getenv("TFFT_STATS_TILE")) c->stats_tile = atoi(e);
    if (const char* e = getenv("TFFT_STATS_PRIO")) c->stats_prio = atoi(e);
    if (const char* e = getenv("TFFT_STATS_FAIL_ONCE")) c->stats_fail_once = atoi(e);
    if (const char* e = getenv("TFFT_STATS_TILE_STEP")) { c->stats_tile_step = atoi(e); if (c->stats_tile_step < 8) c->stats_tile_step = 8; c->stats_tile_step_forced = 1; }
    if (const char* e = getenv("TFFT_COLS_TILES")) { c->cols_tiles_per_block = atoi(e) > 0 ? atoi(e) : 1; c->cols_tiles_forced = 1; }
    if (const char* e = getenv("TFFT_COLS_TILES_EMBED")) c->cols_tiles_embed = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("TFFT_COLS_TILES_STAT")) c->cols_tiles_stat = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("TFFT_COLS_TILES_READ")) c->cols_tiles_read = atoi(e) > 0 ? atoi(e) : 1;
    if (c->cols_direct_max_log > 10) c->cols_direct_max_log = 10;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TFFT_E_HIP; }
    c->own_stream = true;
    (void)hipEventCreate(&c->ev_t0); (void)hipEventCreate(&c->ev_t1);
    c->slots.resize(n_slots);
    const size_t M = (size_t)pw / 2;
    c->slot_stride = 3 * (size_t)ph * M;
    // one slot per value of a plane (PH*(M+1)) + the slack of COLS_STAT's per-wave reservations (TFFT_STAT_RESV slots at a time: < 19 % even
    // when every value is a candidate)
    c->cand_stride = (size_t)ph * (M + 1) + (size_t)ph * M / 4 + 256;
    c->img_stride_b = (((size_t)max_w * max_h * 3 + 255) / 256) * 256;
    const size_t ns = (size_t)n_slots;
    int rc = dev_alloc(c, (void**)&c->img_pool, ns * c->img_stride_b + 256);
    if (!rc) rc = dev_alloc(c, (void**)&c->spec_pool, ns * c->slot_stride * sizeof(float2));
    if (!rc) rc = dev_alloc(c, (void**)&c->tmp_pool, ns * c->slot_stride * sizeof(float2));
    if (!rc) rc = dev_alloc(c, (void**)&c->cand_pool, ns * 3 * c->cand_stride * sizeof(unsigned));
    if (!rc) rc = dev_alloc(c, (void**)&c->col0_pool, ns * 3 * (size_t)ph * sizeof(float2));
    if (!rc) rc = dev_alloc(c, (void**)&c->sel, ns * 3 * sizeof(SelectState));
    if (!rc) rc = dev_alloc(c, (void**)&c->med, ns * 3 * sizeof(float));
    if (!rc) rc = dev_alloc(c, (void**)&c->partial, (ns * 3 * TFFT_STAT_MAX_BLOCKS + ns) * sizeof(unsigned));      // + one flag per image (batch capacity)
    if (!rc) rc = dev_alloc(c, (void**)&c->amb, ns * 3 * TFFT_AMB_CAP * sizeof(float));
    if (!rc) rc = dev_alloc(c, (void**)&c->usable, ns * sizeof(unsigned long long));
    if (!rc) rc = dev_alloc(c, (void**)&c->err, sizeof(int));
    if (!rc) rc = dev_alloc(c, (void**)&c->trash, 8192);
    if (!rc) rc = dev_alloc(c, (void**)&c->last_row, 2 * sizeof(int));
    if (!rc && hipMemset(c->err, 0, sizeof(int)) != hipSuccess) rc = TFFT_E_HIP;
    if (!rc && hipMemset(c->sel, 0, ns * 3 * sizeof(SelectState)) != hipSuccess) rc = TFFT_E_HIP;      // the compact statistics pipeline starts from clean histograms
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = TFFT_E_HIP;
    if (rc != TFFT_OK) { tfft_destroy(c); return rc; }
    *out = c;
    return TFFT_OK;
}

int tfft_destroy(tfft_ctx* c) {
    if (!c) return TFFT_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    invalidate_graphs(c);
    (void)hipFree(c->img_pool); (void)hipFree(c->spec_pool); (void)hipFree(c->tmp_pool); (void)hipFree(c->cand_pool);
    (void)hipFree(c->col0_pool);
    (void)hipFree(c->sel); (void)hipFree(c->med); (void)hipFree(c->partial); (void)hipFree(c->amb); (void)hipFree(c->usable); (void)hipFree(c->err); (void)hipFree(c->ex_cand); (void)hipFree(c->ex_val); (void)hipFree(c->ex_below); (void)hipFree(c->ex_n); for (auto& kv : c->ex_table) (void)hipFree(kv.second); (void)hipFree(c->trash); (void)hipFree(c->bit_index); (void)hipFree(c->last_row);
    for (auto& b : c->tb) { (void)hipFree(b.cnt); (void)hipFree(b.off); (void)hipFree(b.ent); (void)hipFree(b.fl); (void)hipFree(b.pb); }
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    for (auto& kv : c->dc) (void)hipFree(kv.second);
    (void)hipFree(c->stage_bins); (void)hipFree(c->stage_bits); (void)hipFree(c->stage_jit); (void)hipFree(c->stage_out);
    (void)hipFree(c->out_pool); (void)hipFree(c->stream_bits); (void)hipFree(c->stream_plen);
    (void)hipFree(c->sio_hdr); (void)hipFree(c->sio_pay); (void)hipFree(c->sio_status);
    for (int i = 0; i < 4; i++) { if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]); if (c->ev_comp[i]) (void)hipEventDestroy(c->ev_comp[i]); if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]); }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (int i = 0; i < 2; i++) {
        if (c->stream_stats[i]) (void)hipStreamDestroy(c->stream_stats[i]);
        if (c->ev_stats_fork[i]) (void)hipEventDestroy(c->ev_stats_fork[i]);
        if (c->ev_stats_join[i]) (void)hipEventDestroy(c->ev_stats_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TFFT_OK;
}

int tfft_set_stream(tfft_ctx* c, void* hip_stream) {
    if (!c) return TFFT_E_INVALID;
    invalidate_graphs(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return TFFT_OK;
}

int tfft_sync(tfft_ctx* c) {
    if (!c) return TFFT_E_INVALID;
    return check_err_flag(c);
}

int tfft_last_hip_error(const tfft_ctx* c) { return c ? c->last_hip : 0; }

int tfft_plan_info(const tfft_ctx* c, int w, int h, int n_images, int info[4]) {
    if (!c || !info || w < 1 || h < 1 || n_images < 1) return TFFT_E_INVALID;
    int pw = next_pow2(w), ph = next_pow2(h);
    if (pw < 2) pw = 2;
    if (pw > TFFT_MAX_DIM || ph > TFFT_MAX_DIM) return TFFT_E_TOO_LARGE;
    const ColPlan p = plan_cols(c, ph, pw, n_images);
    info[0] = p.direct ? 1 : 0; info[1] = p.log_n1; info[2] = p.log_n2; info[3] = p.fused_fwd ? 1 : 0;
    return TFFT_OK;
}
size_t tfft_device_bytes(const tfft_ctx* c) { return c ? c->dev_bytes : 0; }

int tfft_forward_rgb8_dev(tfft_ctx* c, int slot, const void* rgb_dev, int w, int h, int center, int* pw, int* ph) {
    if (!slot_ok(c, slot) || !rgb_dev) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    int rc = set_geometry(c, s, w, h, center);
    if (rc) return rc;
    if (pw) *pw = s.PW;
    if (ph) *ph = s.PH;
    rc = enqueue_forward(c, slot, 1, (const uint8_t*)rgb_dev, c->stream);
    s.rgb_src = (const uint8_t*)rgb_dev;
    return rc;
}

int tfft_forward_rgb8(tfft_ctx* c, int slot, const uint8_t* rgb, int w, int h, int center, int* pw, int* ph) {
    if (!slot_ok(c, slot) || !rgb) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    int rc = set_geometry(c, s, w, h, center);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->img(slot), rgb, (size_t)w * h * 3, hipMemcpyHostToDevice, c->stream));
    if (pw) *pw = s.PW;
    if (ph) *ph = s.PH;
    rc = enqueue_forward(c, slot, 1, c->img(slot), c->stream);
    s.rgb_src = c->img(slot);
    return rc;
}

// ---- exact statistics of a single resident image (tfft_exact.hip): the bins whose fp32 magnitude lies within a window of the decision
// value are re-evaluated in fp64 from the pixels, everything else is counted on the fp32 spectrum
namespace {
constexpr int EX_CAP = 4096;          // candidate slots per plane
constexpr int EX_SPLIT = 64;          // workgroups (row ranges) a candidate's sum is dealt to, at most

int exact_buffers(tfft_ctx* c) {
    if (c->ex_cand) return TFFT_OK;
    if (dev_alloc(c, (void**)&c->ex_cand, (size_t)3 * EX_CAP * sizeof(ExactCand)) || dev_alloc(c, (void**)&c->ex_val, (size_t)3 * EX_CAP * EX_SPLIT * sizeof(double2)) ||
        dev_alloc(c, (void**)&c->ex_below, 3 * sizeof(unsigned long long)) || dev_alloc(c, (void**)&c->ex_n, 3 * sizeof(unsigned)))
        return TFFT_E_NOMEM;
    return TFFT_OK;
}
bool exact_possible(const tfft_ctx* c, const Slot& s) {
    return c->exact_stats && s.rgb_src && s.PW == s.PWi && s.PWi <= 8192 && s.PH <= 65535 && s.PWi <= 65535;
}
struct ExactOut { std::vector<ExactCand> cand[3]; std::vector<double> mag[3]; unsigned long long outside[3]; };
// one collect + evaluate round with the fp32 |F|^2 windows [lo2, hi2] per plane; false in `ok` when a candidate list overflowed
int exact_round(tfft_ctx* c, int slot, const ExactCollect& P, ExactOut& o, bool& ok) {
    const Slot& s = c->slots[slot];
    int rc = exact_buffers(c);
    if (rc) return rc;
    HIPCHK(c, launch_exact_collect(c->spec(slot), P, c->ex_cand, c->ex_below, c->ex_n, c->stream));
    unsigned n[3];
    HIPCHK(c, hipMemcpyAsync(n, c->ex_n, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(o.outside, c->ex_below, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ok = n[0] <= (unsigned)EX_CAP && n[1] <= (unsigned)EX_CAP && n[2] <= (unsigned)EX_CAP;
    if (!ok) return TFFT_OK;
    double2* table = nullptr;
    {
        auto it = c->ex_table.find(s.PWi);
        if (it == c->ex_table.end()) {
            if (dev_alloc(c, (void**)&table, (size_t)s.PWi * sizeof(double2))) return TFFT_E_NOMEM;
            HIPCHK(c, launch_exact_table(table, s.PWi, c->stream));
            c->ex_table[s.PWi] = table;
        } else table = it->second;
    }
    int split = s.H / 32;
    if (split < 1) split = 1;
    if (split > EX_SPLIT) split = EX_SPLIT;
    for (int p = 0; p < 3; p++) {
        o.cand[p].resize(n[p]); o.mag[p].resize(n[p]);
        if (!n[p]) continue;
        HIPCHK(c, launch_exact_eval(s.rgb_src, s.W, s.H, s.PWi, s.PH, s.center, c->ex_cand + (size_t)p * EX_CAP, n[p], split, table,
                                    c->ex_val + (size_t)p * EX_CAP * EX_SPLIT, c->stream));
    }
    std::vector<double2> v((size_t)EX_CAP * EX_SPLIT);
    for (int p = 0; p < 3; p++) {
        if (!n[p]) continue;
        HIPCHK(c, hipMemcpyAsync(o.cand[p].data(), c->ex_cand + (size_t)p * EX_CAP, n[p] * sizeof(ExactCand), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(v.data(), c->ex_val + (size_t)p * EX_CAP * EX_SPLIT, (size_t)n[p] * split * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (unsigned i = 0; i < n[p]; i++) {
            double re = 0.0, im = 0.0;
            for (int k = 0; k < split; k++) { re += v[(size_t)i * split + k].x; im += v[(size_t)i * split + k].y; }      // fixed order: deterministic
            o.mag[p][i] = hypot(re, im);      // std::abs(complex<double>) of S:406 / S:1004
        }
    }
    return TFFT_OK;
}
// fp32 window around a decision value d (> 0) on |F|^2: relative half-width `rel`, rounded outwards
void exact_window(double d, double rel, float& lo2, float& hi2) {
    const double lo = d * (1.0 - rel), hi = d * (1.0 + rel);
    lo2 = nextafterf((float)(lo * lo), -INFINITY); hi2 = nextafterf((float)(hi * hi), INFINITY);
    if (!(lo2 > 0.f)) lo2 = 0.f;
}
// median_abs S:404-409 for the three planes, refined from the fp32 medians m32; false when the refinement did not apply
bool exact_medians(tfft_ctx* c, int slot, const float m32[3], double med[3], int* rc_out) {
    const Slot& s = c->slots[slot];
    *rc_out = TFFT_OK;
    c->ex_last[0] = c->ex_last[1] = c->ex_last[2] = 0;
    if (!exact_possible(c, s)) return false;
    const unsigned long long rank = ((unsigned long long)s.PH * s.PW) / 2;
    double rel = 2e-6;                   // ~16 sigma of the fp32 transform's error at the median's magnitude (measured 1.2e-7 relative)
    for (int attempt = 0; attempt < 5; attempt++, rel *= 4.0) {
        ExactCollect P{};
        P.PH = s.PH; P.PW = s.PWi; P.PW_full = s.PW; P.cap = 0; P.cap_cand = EX_CAP;
        for (int p = 0; p < 3; p++) exact_window((double)m32[p], rel, P.lo2[p], P.hi2[p]);
        ExactOut o; bool ok = false;
        int rc = exact_round(c, slot, P, o, ok);
        if (rc) { *rc_out = rc; return false; }
        if (!ok) return false;           // a flat spectrum (thousands of bins within 1e-6 of the median): keep the fp32 answer
        bool good = true;
        for (int p = 0; p < 3 && good; p++) {
            const size_t n = o.cand[p].size();
            unsigned long long wsum = 0; double emax = 0.0;
            for (size_t i = 0; i < n; i++) { wsum += o.cand[p][i].w; emax = fmax(emax, fabs(o.mag[p][i] - sqrt((double)o.cand[p][i].m2))); }
            // the window must hold the rank, and be wide against the fp32 error actually seen on its own bins (else a bin outside it
            // could belong inside): 4 x the largest error
            if (!(o.outside[p] <= rank && rank < o.outside[p] + wsum) || 4.0 * emax > rel * (double)m32[p]) { good = false; break; }
            std::vector<size_t> ord(n);
            for (size_t i = 0; i < n; i++) ord[i] = i;
            std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return o.mag[p][a] < o.mag[p][b]; });
            unsigned long long cum = o.outside[p];
            med[p] = (double)m32[p];
            for (size_t k = 0; k < n; k++) { cum += o.cand[p][ord[k]].w; if (rank < cum) { med[p] = o.mag[p][ord[k]]; break; } }
            c->ex_last[p] = (int)n;
        }
        if (good) return true;
    }
    return false;
}
}  // namespace

int tfft_medians(tfft_ctx* c, int slot, double med[3]) {
    if (!slot_ok(c, slot) || !med) return TFFT_E_INVALID;
    if (!c->slots[slot].has_spec) return TFFT_E_STATE;
    int rc = enqueue_medians(c, slot, 1, c->stream);
    if (rc) return rc;
    float m[3];
    HIPCHK(c, hipMemcpyAsync(m, c->med + 3 * slot, sizeof m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) med[i] = (double)m[i];
    // the fp32 medians locate the element; its value (and which of the near-equal neighbours it is) comes from fp64 sums over the pixels
    double ex[3];
    if (exact_medians(c, slot, m, ex, &rc)) { for (int i = 0; i < 3; i++) med[i] = ex[i]; }
    return rc;
}

int tfft_exact_info(const tfft_ctx* c, int n_fp64[3]) {
    if (!c || !n_fp64) return TFFT_E_INVALID;
    for (int i = 0; i < 3; i++) n_fp64[i] = c->ex_last[i];
    return TFFT_OK;
}

int tfft_median_path(tfft_ctx* c, int slot, int fast[3]) {
    if (!slot_ok(c, slot) || !fast) return TFFT_E_INVALID;
    SelectState* h = (SelectState*)malloc(3 * sizeof(SelectState));
    if (!h) return TFFT_E_NOMEM;
    hipError_t e = hipMemcpyAsync(h, c->sel + 3 * slot, 3 * sizeof(SelectState), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int i = 0; i < 3; i++) fast[i] = (h[i].done == 1 && h[i].fast) ? 1 : 0;
    free(h);
    if (e != hipSuccess) { c->last_hip = (int)e; return TFFT_E_HIP; }
    return TFFT_OK;
}

int tfft_capacity(tfft_ctx* c, int slot, double rmin, double rmax, const double thr[3], uint64_t* usable) {
    if (!slot_ok(c, slot) || !thr || !usable) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec) return TFFT_E_STATE;
    CapParams p = cap_params(c, s, rmin, rmax);
    for (int i = 0; i < 3; i++) p.thr[i] = thr[i];
    // exact count (S:998-1008 on the reference's fp64 magnitudes): bins safely above the threshold are counted on the fp32 spectrum, the
    // few within a window of it are settled in fp64.  Small magnitudes carry the transform's ABSOLUTE error (~1e-7 of the spectrum's rms),
    // hence the wide relative window (1e-3) and the check against the error seen on the window's own bins.
    c->ex_last[0] = c->ex_last[1] = c->ex_last[2] = 0;
    if (exact_possible(c, s) && p.bw > 0 && thr[0] > 0.0 && thr[1] > 0.0 && thr[2] > 0.0) {
        double rel = 1e-3;
        for (int attempt = 0; attempt < 4; attempt++, rel *= 4.0) {
            ExactCollect P{};
            P.PH = s.PH; P.PW = s.PWi; P.PW_full = s.PW; P.cap = 1; P.cap_cand = EX_CAP; P.s_lo = p.s_lo; P.s_hi = p.s_hi;
            for (int q = 0; q < 3; q++) exact_window(thr[q], rel, P.lo2[q], P.hi2[q]);
            ExactOut o; bool ok = false;
            int rc = exact_round(c, slot, P, o, ok);
            if (rc) return rc;
            if (!ok) break;
            bool good = true;
            unsigned long long total = 0;
            for (int q = 0; q < 3 && good; q++) {
                unsigned long long cnt = o.outside[q];
                double emax = 0.0;
                for (size_t i = 0; i < o.cand[q].size(); i++) {
                    emax = fmax(emax, fabs(o.mag[q][i] - sqrt((double)o.cand[q][i].m2)));
                    if (!(o.mag[q][i] < thr[q])) cnt += o.cand[q][i].w;          // S:1004: `if (std::abs(F) < thr) continue`
                }
                if (4.0 * emax > rel * thr[q]) good = false;
                total += cnt / 2;                                                  // S:1007: c/2 per plane
                c->ex_last[q] = (int)o.cand[q].size();
            }
            if (good) { *usable = total; return TFFT_OK; }
        }
        c->ex_last[0] = c->ex_last[1] = c->ex_last[2] = 0;
    }
    HIPCHK(c, launch_capacity(c->spec(slot), p, 1, nullptr, c->partial + (size_t)3 * slot * TFFT_STAT_MAX_BLOCKS,
                              c->usable + slot, c->stream));
    unsigned long long u = 0;
    HIPCHK(c, hipMemcpyAsync(&u, c->usable + slot, sizeof u, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *usable = u;
    return TFFT_OK;
}

int tfft_lowfreq_mag(tfft_ctx* c, int slot, int region, double* out) {
    if (!slot_ok(c, slot) || !out || region < 1 || region > 8) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec || !s.rgb_src) return TFFT_E_STATE;
    if (region > s.PH || region > s.PW) return TFFT_E_INVALID;
    // fp64 inner products with the image itself (k_lowfreq_*_f64): scratch = tmp, free between forward and inverse
    const size_t row_bytes = (size_t)s.H * 3 * region * sizeof(double2), out_bytes = (size_t)3 * region * region * sizeof(double);
    if (row_bytes + out_bytes > c->slot_stride * sizeof(float2)) return TFFT_E_INVALID;
    double2* rowsum = (double2*)c->tmp(slot);
    double* d = (double*)((char*)rowsum + row_bytes);
    HIPCHK(c, launch_lowfreq_f64(s.rgb_src, s.W, s.H, s.PW, s.PH, s.center, region, rowsum, d, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, d, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TFFT_OK;
}

int tfft_bins_register_dev(tfft_ctx* c, const void* bins_dev, uint64_t n) {
    if (!c) return TFFT_E_INVALID;
    invalidate_graphs(c);      // a captured sequence may have left the bucket build out
    for (auto& b : c->tb) { b.built_for = nullptr; b.row_for = nullptr; }
    c->reg_bins = (bins_dev && n) ? bins_dev : nullptr;
    c->reg_n = (bins_dev && n) ? n : 0;
    return TFFT_OK;
}

int tfft_set_bit_index(tfft_ctx* c, const uint32_t* bit_index, uint64_t n) {
    if (!c) return TFFT_E_INVALID;
    invalidate_graphs(c);
    for (auto& b : c->tb) { b.built_for = nullptr; b.row_for = nullptr; }
    if (!bit_index || n == 0) {            // back to "bins[i] carries bit i"
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->bit_index);
        c->bit_index = nullptr; c->bit_index_n = 0;
        return TFFT_OK;
    }
    if (n > 0xFFFFFFFFull) return TFFT_E_TOO_LARGE;
    // the kernels index bits/jitter/bits_out with these values: they must be a permutation of 0..n-1
    std::vector<uint64_t> seen((n + 63) / 64, 0);
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t j = bit_index[i];
        if (j >= n || (seen[j >> 6] >> (j & 63)) & 1) return TFFT_E_INVALID;
        seen[j >> 6] |= 1ull << (j & 63);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->bit_index_n != n || !c->bit_index) {
        (void)hipFree(c->bit_index);
        c->bit_index = nullptr; c->bit_index_n = 0;
        int rc = dev_alloc(c, (void**)&c->bit_index, n * sizeof(uint32_t));
        if (rc) return rc;
    }
    HIPCHK(c, hipMemcpy(c->bit_index, bit_index, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->bit_index_n = n;
    return TFFT_OK;
}

int tfft_embed_bins_dev(tfft_ctx* c, int slot, const void* bins, const void* bits, const void* jitter, uint64_t n,
                        double alpha, int adaptive, const double med[3]) {
    if (!slot_ok(c, slot) || (n && (!bins || !bits)) || (adaptive && !med)) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec || !index_ok(c, n)) return TFFT_E_STATE;
    EmbedParams p = embed_params(c, s, n, alpha, adaptive, med, jitter != nullptr);
    HIPCHK(c, launch_embed(c->spec(slot), (const tfft_bin*)bins, (const uint8_t*)bits, (const float*)jitter, p, 1, c->err, c->stream));
    return TFFT_OK;
}

int tfft_embed_bins(tfft_ctx* c, int slot, const tfft_bin* bins, const uint8_t* bits, const float* jitter, uint64_t n,
                    double alpha, int adaptive, const double med[3]) {
    if (!slot_ok(c, slot) || (n && (!bins || !bits))) return TFFT_E_INVALID;
    if (!c->slots[slot].has_spec) return TFFT_E_STATE;
    if (n == 0) return TFFT_OK;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage_bins, bins, n * sizeof(tfft_bin), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->stage_bits, bits, n, hipMemcpyHostToDevice, c->stream));
    if (jitter) HIPCHK(c, hipMemcpyAsync(c->stage_jit, jitter, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = tfft_embed_bins_dev(c, slot, c->stage_bins, c->stage_bits, jitter ? c->stage_jit : nullptr, n, alpha, adaptive, med);
    if (rc) return rc;
    return check_err_flag(c);
}

int tfft_read_bins_dev(tfft_ctx* c, int slot, const void* bins, const void* jitter, uint64_t n, double alpha,
                       int adaptive, const double med[3], void* bits_out) {
    if (!slot_ok(c, slot) || (n && (!bins || !bits_out)) || (adaptive && !med)) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec || !index_ok(c, n)) return TFFT_E_STATE;
    EmbedParams p = embed_params(c, s, n, alpha, adaptive, med, jitter != nullptr);
    HIPCHK(c, launch_read(c->spec(slot), (const tfft_bin*)bins, (const float*)jitter, p, 1, (uint8_t*)bits_out, c->err, c->stream));
    return TFFT_OK;
}

int tfft_read_bins(tfft_ctx* c, int slot, const tfft_bin* bins, const float* jitter, uint64_t n, double alpha,
                   int adaptive, const double med[3], uint8_t* bits_out) {
    if (!slot_ok(c, slot) || (n && (!bins || !bits_out))) return TFFT_E_INVALID;
    if (!c->slots[slot].has_spec) return TFFT_E_STATE;
    if (n == 0) return TFFT_OK;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage_bins, bins, n * sizeof(tfft_bin), hipMemcpyHostToDevice, c->stream));
    if (jitter) HIPCHK(c, hipMemcpyAsync(c->stage_jit, jitter, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = tfft_read_bins_dev(c, slot, c->stage_bins, jitter ? c->stage_jit : nullptr, n, alpha, adaptive, med, c->stage_out);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(bits_out, c->stage_out, n, hipMemcpyDeviceToHost, c->stream));
    return check_err_flag(c);
}

int tfft_inverse_rgb8_dev(tfft_ctx* c, int slot, void* rgb_out_dev) {
    if (!slot_ok(c, slot) || !rgb_out_dev) return TFFT_E_INVALID;
    if (!c->slots[slot].has_spec) return TFFT_E_STATE;
    return enqueue_inverse(c, slot, 1, (uint8_t*)rgb_out_dev, c->stream);
}

int tfft_inverse_rgb8(tfft_ctx* c, int slot, uint8_t* rgb_out) {
    if (!slot_ok(c, slot) || !rgb_out) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec) return TFFT_E_STATE;
    int rc = enqueue_inverse(c, slot, 1, c->img(slot), c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(rgb_out, c->img(slot), (size_t)s.W * s.H * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TFFT_OK;
}

int tfft_download_spectrum(tfft_ctx* c, int slot, float* out) {
    if (!slot_ok(c, slot) || !out) return TFFT_E_INVALID;
    Slot& s = c->slots[slot];
    if (!s.has_spec) return TFFT_E_STATE;
    const size_t n = (size_t)3 * s.PH * s.PW;
    float2* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(float2)) != hipSuccess) return TFFT_E_NOMEM;
    hipError_t e = launch_export_full(c->spec(slot), s.PH, s.PWi, s.PW, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { c->last_hip = (int)e; return TFFT_E_HIP; }
    return TFFT_OK;
}

// ---------------------------------------------------------------- batches
// Images are processed in chunks of n_slots; every stage of a chunk is ONE launch over all its
// images (grid.z = image), so each kernel sees thousands of workgroups.
static int batch_geometry(tfft_ctx* c, int g, int w, int h, int center) {
    for (int i = 0; i < g; i++) { int rc = set_geometry(c, c->slots[i], w, h, center); if (rc) return rc; }
    return TFFT_OK;
}

// device buffers of the tile buckets for `n` bins and `nb` buckets on compute stream `which`
static int ensure_buckets(tfft_ctx* c, int which, uint64_t n, int nb, bool with_values = false) {
    auto& b = c->tb[which];
    if (n > b.cap || !b.ent) {
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        invalidate_graphs(c);
        b.built_for = nullptr;
        (void)hipFree(b.ent); b.ent = nullptr; b.cap = 0;
        const uint64_t cap = n + n / 4 + 1024;
        if (dev_alloc(c, (void**)&b.ent, cap * sizeof(TileBin))) return TFFT_E_NOMEM;
        b.cap = cap;
    }
    if (nb + 1 > b.nb_cap || !b.cnt) {
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        invalidate_graphs(c);
        b.built_for = nullptr;
        (void)hipFree(b.cnt); (void)hipFree(b.off); b.cnt = b.off = nullptr; b.nb_cap = 0;
        if (dev_alloc(c, (void**)&b.cnt, (size_t)(nb + 1) * sizeof(unsigned)) || dev_alloc(c, (void**)&b.off, (size_t)(nb + 1 + (nb + 1023) / 1024) * sizeof(unsigned)))
            return TFFT_E_NOMEM;
        b.nb_cap = nb + 1;
    }
    if (with_values && (n * (uint64_t)c->n_slots > b.fl_cap || !b.fl)) {
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        invalidate_graphs(c);
        (void)hipFree(b.fl); (void)hipFree(b.pb); b.fl = nullptr; b.pb = nullptr; b.fl_cap = 0;
        const uint64_t cap = (n + n / 4 + 1024) * (uint64_t)c->n_slots;
        if (dev_alloc(c, (void**)&b.fl, cap * sizeof(float2)) || dev_alloc(c, (void**)&b.pb, cap)) return TFFT_E_NOMEM;
        b.fl_cap = cap;
    }
    return TFFT_OK;
}

// the bins of a list bucketed by (plane, group, column tile) for the tile-resident read and the delta embedding; a registered list
// (tfft_bins_register_dev) keeps its buckets from one call to the next
static int build_buckets(tfft_ctx* c, int which, const tfft_bin* bins, uint64_t n_bits, const Slot& s, int G, hipStream_t st) {
    auto& tb = c->tb[which];
    const bool registered = bins == c->reg_bins && n_bits == c->reg_n;
    if (!(registered && tb.built_for == bins && tb.built_n == n_bits && tb.built_ph == s.PH && tb.built_pw == s.PWi && tb.built_g == G &&
          tb.built_index == c->bit_index)) {
        // a sequence captured for a registered list holds no bucket build: once the buffers describe another list it must not be replayed
        if (tb.built_for) c->graphs_stale = true;
        HIPCHK(c, launch_bucket_bins(bins, c->bit_index, n_bits, s.PH, s.PWi, G, tb.cnt, tb.off, tb.ent, c->err, c->tile_read == 2, st));
        tb.built_for = registered ? bins : nullptr; tb.built_n = n_bits; tb.built_ph = s.PH; tb.built_pw = s.PWi; tb.built_g = G; tb.built_index = c->bit_index;
        tb.built_bad = false;
        if (registered && st == c->stream) {      // the cached buckets outlive this call: so does the verdict on the list (one sync per registration and geometry)
            const int rc = check_err_flag(c);
            if (rc == TFFT_E_BIN_RANGE) tb.built_bad = true;
            else if (rc) return rc;
        }
    }
    // every call on a registered list with bins outside the grid reports them (at its end, as the call that built the buckets does),
    // not only the first: the flag the builder raised is raised again
    if (tb.built_bad) HIPCHK(c, hipMemsetAsync(c->err, 0x01, sizeof(int), st));
    return TFFT_OK;
}

// one chunk (slots [s0, s0+g), equal geometry) of the two batched pipelines
// forward transform + statistics of slots [s0, s0+g) without a stored spectrum (see ColParams::st_*): em carries the delta-embedding lists
// the statistics' side stream: lowest priority, so that its small kernels fill gaps instead of taking workgroup slots from the transform
// they run beside (TFFT_STATS_PRIO=0: default priority)
static hipError_t create_stats_stream(tfft_ctx* c, hipStream_t* out) {
    int lo = 0, hi = 0;
    if (c->stats_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, lo);      // numerically greatest = lowest priority
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Do the statistics of a batched delta embed run inside the last forward column step (COLS_STAT)?  Two-step column plans with 16 .. 512
// rows per step and whole column tiles; planes up to 2^24 bins (the compact select); an annulus that stays left of column PW/2
// (COLS_STAT counts stored bins only: on tall grids, whose annulus reaches the mirror half, the |F|^2 planes serve); launches of at
// least 2^24 bins -- four 1080p images, one 4K image (TFFT_STATS_TILE=2: any)
static bool tilestats_applies(const tfft_ctx* c, const Slot& s, const ColPlan& pl, const CapParams& p, int n) {
    if (!(c->stats_tile && c->stats_fused && c->stats_compact && !c->median_force_fallback)) return false;
    // a small launch is bound by its dependent launches: this form has 12, the planes' 7 (one 1080p image: 0.314 vs 0.283 ms per round trip)
    if (c->stats_tile < 2 && (unsigned long long)n * s.PH * s.PWi < (1ull << 24)) return false;
    const unsigned long long mirror_d = (unsigned long long)(p.PW - s.PWi / 2) * (unsigned long long)(p.PW - s.PWi / 2);
    return p.bw > 0 && p.s_lo <= p.s_hi && p.s_hi < 0xFFFFFFFFull && mirror_d > p.s_hi && !pl.direct && pl.log_n2 >= 4 && pl.log_n2 <= 9 &&
           (unsigned long long)s.PH * s.PWi <= (1ull << 24) && (s.PWi / 2) % 16 == 0;
}

// The statistics after COLS_STAT, in two parts so that the first can run beside the inverse transform (embed_chunk):
//   select: the medians out of the staged candidates (5 small dependent launches, ~75 us of latency for a 32 x 1080p launch)
//   tail  : images with a plane the fast path could not settle get their spectrum after all -- the plain last forward step, gated (it
//           returns at once for the others) -- then the fallback kernels and the capacities.  Reads `tmp` (the last step's input):
//           the inverse in between keeps its intermediate in `spec` (StageMode::inv_via_spec)
static int enqueue_tilestats_select(tfft_ctx* c, int s0, int g, hipStream_t st) {
    const Slot& s = c->slots[s0];
    HIPCHK(c, launch_stat_select(s.PH, g, c->sel + 3 * s0, c->cand_pool + (size_t)3 * s0 * c->cand_stride, c->cand_stride, c->med + 3 * s0,
                                 c->col0_pool + (size_t)s0 * 3 * s.PH, st));
    return TFFT_OK;
}
static int enqueue_tilestats_tail(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, hipStream_t st, const CapParams& cap, unsigned long long* usable) {
    const Slot& s = c->slots[s0];
    const ColPlan pl = plan_cols(c, s.PH, s.PWi, g);
    ColParams gt{};
    gt.gate = c->sel + 3 * s0;
    StageMode mg; mg.fwd_plain_extra = &gt;
    int rc = enqueue_fft_stage(c, s0, g, pl.direct ? COLS_FWD_A : COLS_FWD_B, rgb_in, nullptr, st, mg);
    if (rc) return rc;
    HIPCHK(c, launch_stat_settle(c->spec(s0), s.PH, s.PWi, c->slot_stride, g, c->sel + 3 * s0, c->med + 3 * s0, &cap,
                                 c->partial + (size_t)s0 * (3 * TFFT_STAT_MAX_BLOCKS + 1), c->amb + (size_t)3 * s0 * TFFT_AMB_CAP, usable, st));
    for (int i = 0; i < g; i++) { c->slots[s0 + i].has_spec = false; c->slots[s0 + i].rgb_src = nullptr; }
    return TFFT_OK;
}

// phases (tfft_profile_stage times them apart): 1 the steps before the last column step, 2 sample + bracket guess, 4 the COLS_STAT step,
// 8 select, 16 gated spectrum + fallbacks + capacity
static int enqueue_forward_tilestats(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, hipStream_t st, ColParams& em, const CapParams& cap,
                                     unsigned long long* usable, int phases = 31) {
    const Slot& s = c->slots[s0];
    const ColPlan pl = plan_cols(c, s.PH, s.PWi, g);
    const int final_fwd = pl.direct ? COLS_FWD_A : COLS_FWD_B;
    // the sample: column tiles off, off + step, ..  -- centred in their strides (tiles 0, step, .. sit at the low-frequency end of every
    // stride and read a median several per cent too high: the bracket missed on every padded image)
    // eight sampled tiles per plane row group (every 8th of a 2048-wide grid's 64, every 16th of a 4096-wide one's 128): the sample pass is a
    // chain of dependent tile transforms per workgroup slot, its time goes with the tiles it walks (8 x 4K: 0.117 -> 0.0x ms)
    const int M = s.PWi / 2, ntiles = (M + 15) / 16;
    int step = c->stats_tile_step;
    while (!c->stats_tile_step_forced && ntiles / step > 8) step *= 2;
    // ... and of those tiles every 4th row group only, twice the tiles instead: a sixteenth of the plane rather than an eighth, spread over twice
    // the columns (the rows of a group are G apart: a regular subsample)
    const int G = 1 << pl.log_n1;
    int g_step = 1;
    if (!c->stats_tile_step_forced && G >= 8 && step >= 2 && ntiles / step >= 2) { g_step = 4; step /= 2; }
    const int off = ntiles > step / 2 ? step / 2 : 0;
    const int Ms = 16 * ((ntiles - off + step - 1) / step);
    int rc;
    for (int stage : {ROWS_FWD, COLS_FWD_A}) {
        if (stage == final_fwd || !(phases & 1)) break;
        rc = enqueue_fft_stage(c, s0, g, stage, rgb_in, nullptr, st);
        if (rc) return rc;
    }
    float2* col0 = c->col0_pool + (size_t)s0 * 3 * s.PH;
    SelectState* sel = c->sel + 3 * s0;
    unsigned* partial = c->partial + (size_t)s0 * (3 * TFFT_STAT_MAX_BLOCKS + 1);
    float* amb = c->amb + (size_t)3 * s0 * TFFT_AMB_CAP;
    unsigned* cand = c->cand_pool + (size_t)3 * s0 * c->cand_stride;
    // (1) every step-th column tile, transformed and dropped into a histogram of |F|^2 (in LDS, ColParams::hist_sel): its median
    // brackets the plane's.  (First form: the tiles written side by side as a narrow spectrum + k_hist_spec over it -- 0.15 ms of a
    // 32 x 1080p launch where this takes 0.0x.)
    ColParams ex{};
    ex.tile_step = step; ex.tile_off = off; ex.out_M = Ms; ex.out_plane_stride = (size_t)s.PH * Ms; ex.out_img_stride = (size_t)3 * s.PH * Ms;
    ex.hist_sel = sel; ex.g_step = g_step; ex.g_off = g_step / 2;
    if (phases & 2) {
        StageMode ms;
        ms.fwd_plain_extra = &ex;
        rc = enqueue_fft_stage(c, s0, g, final_fwd, rgb_in, nullptr, st, ms);
        if (rc) return rc;
        HIPCHK(c, launch_stat_guess(nullptr, s.PH, s.PWi, Ms, ex.out_img_stride, g, sel, &cap, partial, 0, st));
        if (c->stats_skew) HIPCHK(c, launch_skew_bracket(sel, g, c->stats_skew, st));
    }
    // (2) the last forward step: values of the listed bins + the bracket pass on every value
    if (phases & 4) {
    em.st_sel = sel; em.st_cand = cand; em.st_cand_stride = c->cand_stride; em.st_partial = partial; em.st_amb = amb; em.st_col0 = col0;
    em.st_slo = cap.s_lo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)cap.s_lo; em.st_shi = cap.s_hi > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)cap.s_hi;
    em.st_cap = 1; em.st_PW = cap.PW;
    { StageMode me; me.fwd_emit = &em;
      rc = enqueue_fft_stage(c, s0, g, final_fwd, rgb_in, nullptr, st, me); }
    em.st_sel = nullptr;
    if (rc) return rc;
    }
    if (phases & 8) {
        rc = enqueue_tilestats_select(c, s0, g, st);
        if (rc) return rc;
    }
    if (phases & 16) return enqueue_tilestats_tail(c, s0, g, rgb_in, st, cap, usable);
    return TFFT_OK;
}

struct FrameSrc { const uint8_t* hdr; const uint8_t* pay; uint64_t plen; };      // packed frames of a chunk (device), image i at hdr + 38*i / pay + plen*i
static int embed_chunk_impl(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, const tfft_bin* bins, const uint8_t* bits,
                            uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                            unsigned long long* usable, uint8_t* rgb_out, hipStream_t st, uint64_t limit, const FrameSrc* frame);
static int embed_chunk(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, const tfft_bin* bins, const uint8_t* bits,
                       uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                       unsigned long long* usable, uint8_t* rgb_out, hipStream_t st, uint64_t limit = ~0ull, const FrameSrc* frame = nullptr) {
    int rc = usable ? stats_clean_if_dirty(c, st) : TFFT_OK;
    if (!rc) rc = embed_chunk_impl(c, s0, g, rgb_in, bins, bits, n_bits, alpha, rmin, rmax, magmin, usable, rgb_out, st, limit, frame);
    if (!rc && usable && c->stats_fail_once) {      // test hook (TFFT_STATS_FAIL_ONCE): as if the sequence had broken off -- garbage in the select state, an error out
        c->stats_fail_once = 0;
        HIPCHK(c, hipMemsetAsync(c->sel, 0x01, (size_t)c->n_slots * 3 * sizeof(SelectState), st));
        rc = TFFT_E_HIP;
    }
    if (rc && usable) c->stats_dirty = true;      // (the statistics may have been cut off between two of their launches)
    return rc;
}
static int embed_chunk_impl(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, const tfft_bin* bins, const uint8_t* bits,
                            uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                            unsigned long long* usable, uint8_t* rgb_out, hipStream_t st, uint64_t limit, const FrameSrc* frame) {
    const Slot& s = c->slots[s0];
    if (!index_ok(c, n_bits)) return TFFT_E_STATE;
    EmbedParams ep = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
    if (limit < n_bits) ep.limit = limit;      // the stream is shorter than the bin list (image i's bits still n_bits apart)
    if (frame) { ep.frame_hdr = frame->hdr; ep.frame_pay = frame->pay; ep.frame_plen = frame->plen; }
    // Delta embedding.  The inverse transform is linear and IFFT(F) is the cover itself, so the stego image is cover + IFFT(F' - F),
    // and F' - F is zero but for the bins of the list.  The bins are bucketed by column tile (the buckets extraction uses); the last
    // forward column step, which has every tile in LDS, writes the values of the listed bins out in bucket order; the first inverse
    // column step builds its tiles from that list instead of reading the spectrum; nothing writes F' anywhere, and the row kernel adds
    // its result to the cover's pixels.  (The bucket build is per call unless the list is registered, tfft_bins_register_dev.)
    const bool delta = c->embed_delta && !ep.generic && n_bits > 0;      // whatever the chunk size: the bytes of a stego image do not depend on how the batch was cut
    const int which = (c->stream2 && st == c->stream2) ? 1 : 0;
    ColParams em{};
    int rc;
    if (delta) {
        const ColPlan pl = plan_cols(c, s.PH, s.PWi, g);
        const int G = pl.direct ? 1 : (1 << pl.log_n1), ntiles = (s.PWi / 2 + 15) / 16, nb = 3 * ntiles * G;
        rc = ensure_buckets(c, which, n_bits, nb, true);
        if (rc) return rc;
        rc = build_buckets(c, which, bins, n_bits, s, G, st);
        if (rc) return rc;
        auto& tb = c->tb[which];
        em.rd_bins = tb.ent; em.rd_off = tb.off; em.trash = c->trash; em.em_fl = tb.fl + (size_t)s0 * n_bits; em.em_pb = tb.pb + (size_t)s0 * n_bits;
        em.em_n = n_bits; em.em_cos = ep.cos_a; em.em_sin = ep.sin_a;
        if (usable) {
            CapParams p0 = cap_params(c, s, rmin, rmax);
            if (stats_m2_applies(c, s, p0)) { em.em_m2 = 1; em.st_col0 = c->col0_pool + (size_t)s0 * 3 * s.PH; }
        } else if (c->stats_m2) {      // no capacity asked for: nobody reads the spectrum, the last forward step stores nothing
            em.em_m2 = 2; em.st_col0 = c->col0_pool + (size_t)s0 * 3 * s.PH;
        }
        // the stream bits in bucket order (the packed frames of the stream pipelines are expanded on the way).  (On the side stream
        // beside the forward transform it gained nothing measurable: 0.03 ms of 3.3.)
        HIPCHK(c, launch_gather_bits(tb.ent, tb.off + nb, bits, ep.frame_hdr, ep.frame_pay, ep.frame_plen, n_bits, ep.limit, g, tb.pb + (size_t)s0 * n_bits, st));
    }
    if (delta && usable) {
        // the statistics' bracket pass inside the last forward column step: neither the spectrum nor |F|^2 is stored (unless a plane's
        // bracket turns out wrong: then the gated plain step produces the spectrum for the fallback kernels)
        CapParams p = cap_params(c, s, rmin, rmax);
        p.magmin = magmin;
        if (tilestats_applies(c, s, plan_cols(c, s.PH, s.PWi, g), p, g)) {
            em.em_m2 = 0;
            rc = enqueue_forward_tilestats(c, s0, g, rgb_in, st, em, p, usable, 7);
            if (rc) return rc;
            // the select chain is five small dependent launches: on a side stream beside the inverse transform, which does not wait for it
            hipStream_t sst = st;
            if (c->stats_async) {
                if (!c->stream_stats[which]) {
                    HIPCHK(c, create_stats_stream(c, &c->stream_stats[which]));
                    HIPCHK(c, hipEventCreateWithFlags(&c->ev_stats_fork[which], hipEventDisableTiming));
                    HIPCHK(c, hipEventCreateWithFlags(&c->ev_stats_join[which], hipEventDisableTiming));
                }
                sst = c->stream_stats[which];
                HIPCHK(c, hipEventRecord(c->ev_stats_fork[which], st));
                HIPCHK(c, hipStreamWaitEvent(sst, c->ev_stats_fork[which], 0));
            }
            rc = enqueue_tilestats_select(c, s0, g, sst);
            if (rc) return rc;
            StageMode mi;
            mi.inv_embed = &em; mi.inv_cover = rgb_in; mi.inv_via_spec = true;
            rc = enqueue_inverse(c, s0, g, rgb_out, st, mi);
            if (rc) return rc;
            if (sst != st) {
                HIPCHK(c, hipEventRecord(c->ev_stats_join[which], sst));
                HIPCHK(c, hipStreamWaitEvent(st, c->ev_stats_join[which], 0));
            }
            return enqueue_tilestats_tail(c, s0, g, rgb_in, st, p, usable);
        }
    }
    StageMode md;
    if (delta) md.fwd_emit = &em;
    rc = enqueue_forward(c, s0, g, rgb_in, st, md);
    if (rc) return rc;
    hipStream_t sst = st;       // the stream the statistics run on
    const bool async = delta && usable && c->stats_async;
    if (async) {
        if (!c->stream_stats[which]) {
            HIPCHK(c, create_stats_stream(c, &c->stream_stats[which]));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_stats_fork[which], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_stats_join[which], hipEventDisableTiming));
        }
        sst = c->stream_stats[which];
        HIPCHK(c, hipEventRecord(c->ev_stats_fork[which], st));
        HIPCHK(c, hipStreamWaitEvent(sst, c->ev_stats_fork[which], 0));
    }
    if (usable) {      // S:922-923, S:998-1012 on the device, no host round trip: capacity is counted inside the median's full pass
        CapParams p = cap_params(c, s, rmin, rmax);
        p.magmin = magmin;
        if (c->stats_fused && p.bw > 0) {
            rc = enqueue_medians(c, s0, g, sst, &p, usable, em.em_m2 != 0);
            if (rc) return rc;
        } else {
            rc = enqueue_medians(c, s0, g, sst);
            if (rc) return rc;
            HIPCHK(c, launch_capacity(c->spec(s0), p, g, c->med + 3 * s0, c->partial + (size_t)s0 * (3 * TFFT_STAT_MAX_BLOCKS + 1), usable, sst));
        }
    }
    if (delta) {
        StageMode mi;
        mi.inv_embed = &em; mi.inv_cover = rgb_in;
        rc = enqueue_inverse(c, s0, g, rgb_out, st, mi);
        if (async) {            // whoever waits for the context's stream has the capacities too
            HIPCHK(c, hipEventRecord(c->ev_stats_join[which], sst));
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_stats_join[which], 0));
        }
        return rc;
    }
    HIPCHK(c, launch_embed(c->spec(s0), bins, bits, nullptr, ep, g, c->err, st));
    return enqueue_inverse(c, s0, g, rgb_out, st);
}
static int extract_chunk(tfft_ctx* c, int s0, int g, const uint8_t* rgb_in, const tfft_bin* bins, uint64_t n_bits,
                         double alpha, uint8_t* bits_out, hipStream_t st) {
    const Slot& s = c->slots[s0];
    if (!index_ok(c, n_bits)) return TFFT_E_STATE;
    const int which = (c->stream2 && st == c->stream2) ? 1 : 0;
    EmbedParams ep = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
    int rc;
    // (the bucket build is per call: it pays off from about 8 images per chunk; TFFT_TILE_READ=2/3 force it)
    // a registered list keeps its buckets (no per-call build to pay for): then the tile-resident read also serves small chunks of LARGE
    // images (one 4K image: 0.739 -> 0.725 ms per round trip; one 1080p image has too few tiles to fill the chip: 0.265 -> 0.301)
    const bool reg_large = bins == c->reg_bins && n_bits == c->reg_n && (unsigned long long)s.PH * s.PWi >= (1ull << 23);
    // (an alpha outside (0, pi) takes the general phase comparison of k_read: the tile kernel reads the sign of Im only)
    if (c->tile_read && n_bits > 0 && !ep.generic && (g >= 8 || c->tile_read >= 2 || reg_large)) {
        // The spectrum is only ever read at the bins of the list: bucket them by column tile and let the final
        // forward column step read the bits out of its LDS-resident tiles -- no spectrum store, no k_read.
        const ColPlan pl = plan_cols(c, s.PH, s.PWi, g);
        const int G = pl.direct ? 1 : (1 << pl.log_n1), ntiles = (s.PWi / 2 + 15) / 16, nb = 3 * ntiles * G;
        rc = ensure_buckets(c, which, n_bits, nb);
        if (rc) return rc;
        auto& tb = c->tb[which];
        HIPCHK(c, hipMemsetAsync(bits_out, 0, (size_t)g * n_bits, st));          // bins the walk would never produce read as 0 (k_read does the same)
        rc = build_buckets(c, which, bins, n_bits, s, G, st);
        if (rc) return rc;
        ColParams rd{};
        rd.rd_bins = tb.ent; rd.rd_off = tb.off; rd.rd_bits = bits_out; rd.rd_n = n_bits; rd.trash = c->trash;
        StageMode md;
        md.fwd_read = &rd;
        rc = enqueue_forward(c, s0, g, rgb_in, st, md);
        if (rc) return rc;
    } else {
        // the spectrum is only read at the bins of the list: rows above the highest one are never stored
        int* last_row = c->last_row + which;
        auto& tbr = c->tb[which];
        const bool registered = bins == c->reg_bins && n_bits == c->reg_n;
        if (!(registered && tbr.row_for == bins && tbr.row_n == n_bits && tbr.row_ph == s.PH && tbr.row_pw == s.PWi)) {
            HIPCHK(c, launch_bins_last_row(bins, n_bits, s.PH, s.PWi, last_row, st));
            tbr.row_for = registered ? bins : nullptr; tbr.row_n = n_bits; tbr.row_ph = s.PH; tbr.row_pw = s.PWi;
        }
        StageMode md;
        md.fwd_last_row = last_row;
        rc = enqueue_forward(c, s0, g, rgb_in, st, md);
        if (rc) return rc;
        HIPCHK(c, launch_read(c->spec(s0), bins, nullptr, ep, g, bits_out, c->err, st));
    }
    for (int i = 0; i < g; i++) c->slots[s0 + i].has_spec = false;      // partial or no spectrum: not for tfft_medians & co
    return TFFT_OK;
}

// TFFT_STREAMS=2: a chunk of >= 8 images is split in two halves that run on two HIP streams, so that the
// latency-bound kernels of one half overlap the bandwidth-bound kernels of the other.  Returns the size of
// the first half (0: no split) after forking stream2 off the context stream.
static int split_fork(tfft_ctx* c, int g, int* h1) {
    *h1 = 0;
    if (c->n_streams < 2 || g < 8) return TFFT_OK;
    if (!c->stream2) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    *h1 = g / 2;
    return TFFT_OK;
}
static int split_join(tfft_ctx* c) {
    HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return TFFT_OK;
}

static int embed_batch_dev_impl(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                const void* bits_dev, uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                                void* usable_out_dev, void* rgb_out_dev) {
    const size_t img_bytes = (size_t)w * h * 3;
    for (int i0 = 0; i0 < n_images; i0 += c->n_slots) {
        const int g = (n_images - i0 < c->n_slots) ? n_images - i0 : c->n_slots;
        int rc = batch_geometry(c, g, w, h, center);
        if (rc) return rc;
        int h1 = 0;
        rc = split_fork(c, g, &h1);
        if (rc) return rc;
        for (int part = 0; part < (h1 ? 2 : 1); part++) {
            const int s0 = part ? h1 : 0, gp = h1 ? (part ? g - h1 : h1) : g, i1 = i0 + s0;
            rc = embed_chunk(c, s0, gp, (const uint8_t*)rgb_dev + (size_t)i1 * img_bytes, (const tfft_bin*)bins_dev,
                             (const uint8_t*)bits_dev + (size_t)i1 * n_bits, n_bits, alpha, rmin, rmax, magmin,
                             usable_out_dev ? (unsigned long long*)usable_out_dev + i1 : nullptr,
                             (uint8_t*)rgb_out_dev + (size_t)i1 * img_bytes, part ? c->stream2 : c->stream);
            if (rc) return rc;
        }
        if (h1) { rc = split_join(c); if (rc) return rc; }
    }
    return TFFT_OK;
}

// host-side state a batch call leaves behind (what a graph replay has to redo): geometry set, no spectrum in any slot
static int batch_after(tfft_ctx* c, int n_images, int w, int h, int center) {
    const int g = n_images < c->n_slots ? n_images : c->n_slots;
    int rc = batch_geometry(c, g, w, h, center);
    if (rc) return rc;
    for (int i = 0; i < g; i++) { c->slots[i].has_spec = false; c->slots[i].rgb_src = nullptr; }
    return TFFT_OK;
}

int tfft_embed_batch_dev(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                         const void* bits_dev, uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                         void* usable_out_dev, void* rgb_out_dev) {
    if (!c || n_images < 0 || !rgb_dev || !rgb_out_dev || (n_bits && (!bins_dev || !bits_dev))) return TFFT_E_INVALID;
    if (!index_ok(c, n_bits)) return TFFT_E_STATE;
    const std::vector<uint64_t> key = {1, (uint64_t)n_images, key_bits(rgb_dev), (uint64_t)w, (uint64_t)h, (uint64_t)center, key_bits(bins_dev), key_bits(bits_dev),
                                       n_bits, key_bits(alpha), key_bits(rmin), key_bits(rmax), key_bits(magmin), key_bits(usable_out_dev),
                                       key_bits(rgb_out_dev), key_bits(c->bit_index), key_bits((double)c->dc_bias)};
    return with_graph(c, n_images, key,
                      [&] { return embed_batch_dev_impl(c, n_images, rgb_dev, w, h, center, bins_dev, bits_dev, n_bits, alpha, rmin, rmax, magmin, usable_out_dev, rgb_out_dev); },
                      [&] { return batch_after(c, n_images, w, h, center); });
}

static int extract_batch_dev_impl(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                  uint64_t n_bits, double alpha, void* bits_out_dev) {
    const size_t img_bytes = (size_t)w * h * 3;
    for (int i0 = 0; i0 < n_images; i0 += c->n_slots) {
        const int g = (n_images - i0 < c->n_slots) ? n_images - i0 : c->n_slots;
        int rc = batch_geometry(c, g, w, h, center);
        if (rc) return rc;
        int h1 = 0;
        rc = split_fork(c, g, &h1);
        if (rc) return rc;
        for (int part = 0; part < (h1 ? 2 : 1); part++) {
            const int s0 = part ? h1 : 0, gp = h1 ? (part ? g - h1 : h1) : g, i1 = i0 + s0;
            rc = extract_chunk(c, s0, gp, (const uint8_t*)rgb_dev + (size_t)i1 * img_bytes, (const tfft_bin*)bins_dev, n_bits, alpha,
                               (uint8_t*)bits_out_dev + (size_t)i1 * n_bits, part ? c->stream2 : c->stream);
            if (rc) return rc;
        }
        if (h1) { rc = split_join(c); if (rc) return rc; }
    }
    return TFFT_OK;
}

// the generic read path stages its parameter block with a host -> device copy per call: not captured
static bool read_is_simple(double alpha) { return alpha > 0.0 && alpha < M_PI; }

int tfft_extract_batch_dev(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                           uint64_t n_bits, double alpha, void* bits_out_dev) {
    if (!c || n_images < 0 || !rgb_dev || (n_bits && (!bins_dev || !bits_out_dev))) return TFFT_E_INVALID;
    if (!index_ok(c, n_bits)) return TFFT_E_STATE;
    const std::vector<uint64_t> key = {2, (uint64_t)n_images, key_bits(rgb_dev), (uint64_t)w, (uint64_t)h, (uint64_t)center, key_bits(bins_dev), n_bits,
                                       key_bits(alpha), key_bits(bits_out_dev), key_bits(c->bit_index), key_bits((double)c->dc_bias)};
    return with_graph(c, read_is_simple(alpha) ? n_images : 0, key,
                      [&] { return extract_batch_dev_impl(c, n_images, rgb_dev, w, h, center, bins_dev, n_bits, alpha, bits_out_dev); },
                      [&] { return batch_after(c, n_images, w, h, center); });
}

// ---------------------------------------------------------------- packed-byte streams (SURVEY 8 f-3 wired into the pipelines)
static int ensure_stream(tfft_ctx* c, uint64_t n_bins) {
    const size_t need = (size_t)c->n_slots * n_bins;
    if (need <= c->stream_cap && c->stream_plen) return TFFT_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    invalidate_graphs(c);
    (void)hipFree(c->stream_bits); c->stream_bits = nullptr; c->stream_cap = 0;
    if (dev_alloc(c, (void**)&c->stream_bits, need + 64)) return TFFT_E_NOMEM;
    c->stream_cap = need;
    if (!c->stream_plen && dev_alloc(c, (void**)&c->stream_plen, (size_t)c->n_slots * sizeof(unsigned))) return TFFT_E_NOMEM;
    return TFFT_OK;
}

static int embed_stream_batch_dev_impl(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                       uint64_t n_bins, const void* header_dev, const void* payload_dev, uint64_t payload_len, double alpha,
                                       double rmin, double rmax, double magmin, void* usable_out_dev, void* rgb_out_dev) {
    const uint64_t n_bits = 38ull * 24 + payload_len * 56;           // S:986-995
    int rc = TFFT_OK;
    const size_t img_bytes = (size_t)w * h * 3;
    for (int i0 = 0; i0 < n_images; i0 += c->n_slots) {
        const int g = (n_images - i0 < c->n_slots) ? n_images - i0 : c->n_slots;
        rc = batch_geometry(c, g, w, h, center);
        if (rc) return rc;
        // bits_from_bytes + rep3/rep7_encode happen inside k_embed: every bin computes its own stream bit from the packed frame
        const FrameSrc fr{(const uint8_t*)header_dev + (size_t)i0 * 38, (const uint8_t*)payload_dev + (size_t)i0 * payload_len, payload_len};
        rc = embed_chunk(c, 0, g, (const uint8_t*)rgb_dev + (size_t)i0 * img_bytes, (const tfft_bin*)bins_dev, nullptr, n_bins, alpha, rmin, rmax,
                         magmin, usable_out_dev ? (unsigned long long*)usable_out_dev + i0 : nullptr, (uint8_t*)rgb_out_dev + (size_t)i0 * img_bytes,
                         c->stream, n_bits, &fr);
        if (rc) return rc;
    }
    return TFFT_OK;
}

int tfft_embed_stream_batch_dev(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                uint64_t n_bins, const void* header_dev, const void* payload_dev, uint64_t payload_len, double alpha,
                                double rmin, double rmax, double magmin, void* usable_out_dev, void* rgb_out_dev) {
    if (!c || n_images < 0 || !rgb_dev || !rgb_out_dev || !bins_dev || !header_dev || (payload_len && !payload_dev)) return TFFT_E_INVALID;
    if (n_bins < 912 || payload_len > (n_bins - 912) / 56) return TFFT_E_INVALID;      // (before the multiplication: a huge length must not wrap)
    const uint64_t n_bits = 38ull * 24 + payload_len * 56;           // S:986-995
    if (n_bits > n_bins) return TFFT_E_INVALID;                      // the caller's walk is shorter than the stream
    if (!index_ok(c, n_bins)) return TFFT_E_STATE;
    int rc = ensure_stream(c, n_bins);                                // (may reallocate: before any cached sequence is looked up)
    if (rc) return rc;
    const std::vector<uint64_t> key = {3, (uint64_t)n_images, key_bits(rgb_dev), (uint64_t)w, (uint64_t)h, (uint64_t)center, key_bits(bins_dev), n_bins,
                                       key_bits(header_dev), key_bits(payload_dev), payload_len, key_bits(alpha), key_bits(rmin), key_bits(rmax),
                                       key_bits(magmin), key_bits(usable_out_dev), key_bits(rgb_out_dev), key_bits(c->bit_index), key_bits((double)c->dc_bias)};
    return with_graph(c, n_images, key,
                      [&] { return embed_stream_batch_dev_impl(c, n_images, rgb_dev, w, h, center, bins_dev, n_bins, header_dev, payload_dev, payload_len, alpha,
                                                               rmin, rmax, magmin, usable_out_dev, rgb_out_dev); },
                      [&] { return batch_after(c, n_images, w, h, center); });
}

static int extract_stream_batch_dev_impl(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                         uint64_t n_bins, double alpha, void* header_out_dev, void* payload_out_dev, uint64_t max_payload_len,
                                         void* status_out_dev, void* raw_bits_out_dev) {
    int rc = TFFT_OK;
    const size_t img_bytes = (size_t)w * h * 3;
    for (int i0 = 0; i0 < n_images; i0 += c->n_slots) {
        const int g = (n_images - i0 < c->n_slots) ? n_images - i0 : c->n_slots;
        rc = batch_geometry(c, g, w, h, center);
        if (rc) return rc;
        // every position of the caller's walk is read in the one pass that has the spectrum on chip; the header decides
        // afterwards how many of them belong to the stream (S:1223-1264: 912 bits, clen, then 56*(clen+16) more)
        uint8_t* raw = raw_bits_out_dev ? (uint8_t*)raw_bits_out_dev + (size_t)i0 * n_bins : c->stream_bits;
        rc = extract_chunk(c, 0, g, (const uint8_t*)rgb_dev + (size_t)i0 * img_bytes, (const tfft_bin*)bins_dev, n_bins, alpha, raw, c->stream);
        if (rc) return rc;
        HIPCHK(c, launch_stream_decode(raw, n_bins, max_payload_len, g, (uint8_t*)header_out_dev + (size_t)i0 * 38,
                                       (uint8_t*)payload_out_dev + (size_t)i0 * max_payload_len, (int*)status_out_dev + i0, c->stream_plen, c->stream));
    }
    return TFFT_OK;
}

int tfft_extract_stream_batch_dev(tfft_ctx* c, int n_images, const void* rgb_dev, int w, int h, int center, const void* bins_dev,
                                  uint64_t n_bins, double alpha, void* header_out_dev, void* payload_out_dev, uint64_t max_payload_len,
                                  void* status_out_dev, void* raw_bits_out_dev) {
    if (!c || n_images < 0 || !rgb_dev || !bins_dev || n_bins == 0 || !header_out_dev || !status_out_dev || (max_payload_len && !payload_out_dev))
        return TFFT_E_INVALID;
    if (!index_ok(c, n_bins)) return TFFT_E_STATE;
    int rc = TFFT_OK;
    if (!raw_bits_out_dev) { rc = ensure_stream(c, n_bins); if (rc) return rc; }
    else if (!c->stream_plen && dev_alloc(c, (void**)&c->stream_plen, (size_t)c->n_slots * sizeof(unsigned))) return TFFT_E_NOMEM;
    const std::vector<uint64_t> key = {4, (uint64_t)n_images, key_bits(rgb_dev), (uint64_t)w, (uint64_t)h, (uint64_t)center, key_bits(bins_dev), n_bins,
                                       key_bits(alpha), key_bits(header_out_dev), key_bits(payload_out_dev), max_payload_len, key_bits(status_out_dev),
                                       key_bits(raw_bits_out_dev), key_bits(c->bit_index), key_bits((double)c->dc_bias)};
    return with_graph(c, read_is_simple(alpha) ? n_images : 0, key,
                      [&] { return extract_stream_batch_dev_impl(c, n_images, rgb_dev, w, h, center, bins_dev, n_bins, alpha, header_out_dev, payload_out_dev,
                                                                 max_payload_len, status_out_dev, raw_bits_out_dev); },
                      [&] { return batch_after(c, n_images, w, h, center); });
}

// ---------------------------------------------------------------- host-buffer batches
// The slots are split into a ring of up to four parts; while one part computes, the next parts' inputs
// arrive over PCIe on a copy-in stream and earlier results leave on a copy-out stream.  Overlap needs pinned
// host memory (tfft_host_alloc or any page-locked buffer); pageable buffers work but serialise.
static int pipe_init(tfft_ctx* c) {
    if (c->s_in) return TFFT_OK;
    HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_comp[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
    }
    return dev_alloc(c, (void**)&c->out_pool, (size_t)c->n_slots * c->img_stride_b + 256);
}

// packed-byte framing around the host pipeline (tfft_*_stream_batch): header/payload bytes cross PCIe instead of one byte per bit
struct StreamIO {
    const uint8_t* header_in = nullptr; const uint8_t* payload_in = nullptr; uint64_t plen = 0;      // embed
    uint8_t* header_out = nullptr; uint8_t* payload_out = nullptr; uint64_t max_plen = 0; int32_t* status_out = nullptr;   // extract
};
static int ensure_stream_io(tfft_ctx* c, uint64_t plen) {
    if (c->sio_hdr && plen <= c->sio_plen) return TFFT_OK;
    (void)hipStreamSynchronize(c->stream);
    invalidate_graphs(c);
    (void)hipFree(c->sio_hdr); (void)hipFree(c->sio_pay); (void)hipFree(c->sio_status);
    c->sio_hdr = c->sio_pay = nullptr; c->sio_status = nullptr; c->sio_plen = 0;
    const uint64_t cap = plen + plen / 4 + 64;
    if (dev_alloc(c, (void**)&c->sio_hdr, (size_t)c->n_slots * 38) || dev_alloc(c, (void**)&c->sio_pay, (size_t)c->n_slots * cap) ||
        dev_alloc(c, (void**)&c->sio_status, (size_t)c->n_slots * sizeof(int))) return TFFT_E_NOMEM;
    c->sio_plen = cap;
    return TFFT_OK;
}

static int batch_host(tfft_ctx* c, bool embed, int n_images, const uint8_t* rgb, int w, int h, int center,
                      const tfft_bin* bins, const uint8_t* bits, uint64_t n_bits, double alpha, double rmin, double rmax,
                      double magmin, uint64_t* usable, uint8_t* rgb_out, uint8_t* bits_out, const StreamIO* sio = nullptr) {
    if (n_images == 0) return TFFT_OK;
    int rc = pipe_init(c);
    if (rc) return rc;
    if (sio) {
        rc = ensure_stream_io(c, embed ? sio->plen : sio->max_plen);
        if (rc) return rc;
        if (!c->stream_plen && dev_alloc(c, (void**)&c->stream_plen, (size_t)c->n_slots * sizeof(unsigned))) return TFFT_E_NOMEM;
    }
    rc = ensure_stage(c, (uint64_t)c->n_slots * n_bits > n_bits ? (uint64_t)c->n_slots * n_bits : n_bits);
    if (rc) return rc;
    rc = batch_geometry(c, c->n_slots, w, h, center);
    if (rc) return rc;
    { const float2* t; rc = get_twiddles(c, c->slots[0].PWi, &t); if (rc) return rc; rc = get_twiddles(c, c->slots[0].PH, &t); if (rc) return rc; }
    const size_t img_bytes = (size_t)w * h * 3;
    // the slots form a ring of up to four parts: copy-in of part k+1..k+3 overlaps the kernels of part k
    const int nhalves = c->n_slots >= 8 ? 4 : (c->n_slots >= 2 ? 2 : 1);
    const int half = c->n_slots / nhalves;
    HIPCHK(c, hipMemcpyAsync(c->stage_bins, bins, n_bits * sizeof(tfft_bin), hipMemcpyHostToDevice, c->stream));
    int chunk = 0;
    for (int i0 = 0; i0 < n_images; i0 += half, chunk++) {
        const int g = (n_images - i0 < half) ? n_images - i0 : half;
        const int hh = chunk % nhalves, s0 = hh * half;
        uint8_t* d_bits = (uint8_t*)c->stage_bits + (size_t)s0 * n_bits;
        uint8_t* d_bout = (uint8_t*)c->stage_out + (size_t)s0 * n_bits;
        // copy-in: the half's input buffers are free once the chunk that used them has been computed
        HIPCHK(c, hipStreamWaitEvent(c->s_in, c->ev_comp[hh], 0));
        // the pipeline treats the half's staging area as one packed batch buffer (g images back to back)
        HIPCHK(c, hipMemcpyAsync(c->img(s0), rgb + (size_t)i0 * img_bytes, (size_t)g * img_bytes, hipMemcpyHostToDevice, c->s_in));
        if (embed && !sio) HIPCHK(c, hipMemcpyAsync(d_bits, bits + (size_t)i0 * n_bits, (size_t)g * n_bits, hipMemcpyHostToDevice, c->s_in));
        if (embed && sio) {      // 38 + plen bytes per image instead of 912 + 56*plen
            HIPCHK(c, hipMemcpyAsync(c->sio_hdr + (size_t)s0 * 38, sio->header_in + (size_t)i0 * 38, (size_t)g * 38, hipMemcpyHostToDevice, c->s_in));
            if (sio->plen) HIPCHK(c, hipMemcpyAsync(c->sio_pay + (size_t)s0 * sio->plen, sio->payload_in + (size_t)i0 * sio->plen, (size_t)g * sio->plen, hipMemcpyHostToDevice, c->s_in));
        }
        HIPCHK(c, hipEventRecord(c->ev_in[hh], c->s_in));
        // compute: needs the inputs, and the half's output buffers drained by the copy-out of two chunks ago
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in[hh], 0));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_out[hh], 0));
        uint64_t limit = ~0ull;
        FrameSrc fr{nullptr, nullptr, 0};
        if (embed && sio) {      // bits_from_bytes + rep3/rep7 inside k_embed, from the packed frames of this part of the ring
            limit = 38ull * 24 + sio->plen * 56;
            fr = FrameSrc{c->sio_hdr + (size_t)s0 * 38, c->sio_pay + (size_t)s0 * sio->plen, sio->plen};
        }
        if (embed)
            rc = embed_chunk(c, s0, g, c->img(s0), (const tfft_bin*)c->stage_bins, sio ? nullptr : d_bits, n_bits, alpha, rmin, rmax, magmin,
                             usable ? c->usable + s0 : nullptr, c->out_pool + (size_t)s0 * c->img_stride_b, c->stream, limit, sio ? &fr : nullptr);   // packed, like the input
        else
            rc = extract_chunk(c, s0, g, c->img(s0), (const tfft_bin*)c->stage_bins, n_bits, alpha, d_bout, c->stream);
        if (rc) return rc;
        if (!embed && sio)       // header -> clen -> payload on the device: only packed bytes and a status word go back
            HIPCHK(c, launch_stream_decode(d_bout, n_bits, sio->max_plen, g, c->sio_hdr + (size_t)s0 * 38, c->sio_pay + (size_t)s0 * sio->max_plen,
                                           c->sio_status + s0, c->stream_plen + s0, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_comp[hh], c->stream));
        // copy-out
        HIPCHK(c, hipStreamWaitEvent(c->s_out, c->ev_comp[hh], 0));
        if (embed) {
            HIPCHK(c, hipMemcpyAsync(rgb_out + (size_t)i0 * img_bytes, c->out_pool + (size_t)s0 * c->img_stride_b, (size_t)g * img_bytes, hipMemcpyDeviceToHost, c->s_out));
            if (usable) HIPCHK(c, hipMemcpyAsync(usable + i0, c->usable + s0, (size_t)g * sizeof(uint64_t), hipMemcpyDeviceToHost, c->s_out));
        } else if (sio) {
            HIPCHK(c, hipMemcpyAsync(sio->header_out + (size_t)i0 * 38, c->sio_hdr + (size_t)s0 * 38, (size_t)g * 38, hipMemcpyDeviceToHost, c->s_out));
            if (sio->max_plen) HIPCHK(c, hipMemcpyAsync(sio->payload_out + (size_t)i0 * sio->max_plen, c->sio_pay + (size_t)s0 * sio->max_plen, (size_t)g * sio->max_plen, hipMemcpyDeviceToHost, c->s_out));
            HIPCHK(c, hipMemcpyAsync(sio->status_out + i0, c->sio_status + s0, (size_t)g * sizeof(int32_t), hipMemcpyDeviceToHost, c->s_out));
            if (bits_out) HIPCHK(c, hipMemcpyAsync(bits_out + (size_t)i0 * n_bits, d_bout, (size_t)g * n_bits, hipMemcpyDeviceToHost, c->s_out));
        } else {
            HIPCHK(c, hipMemcpyAsync(bits_out + (size_t)i0 * n_bits, d_bout, (size_t)g * n_bits, hipMemcpyDeviceToHost, c->s_out));
        }
        HIPCHK(c, hipEventRecord(c->ev_out[hh], c->s_out));
    }
    HIPCHK(c, hipStreamSynchronize(c->s_out));
    return check_err_flag(c);
}

int tfft_embed_batch(tfft_ctx* c, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                     const uint8_t* bits, uint64_t n_bits, double alpha, double rmin, double rmax, double magmin,
                     uint64_t* usable_out, uint8_t* rgb_out) {
    if (!c || n_images < 0 || !rgb || !rgb_out || !bins || !bits || n_bits == 0) return TFFT_E_INVALID;
    return batch_host(c, true, n_images, rgb, w, h, center, bins, bits, n_bits, alpha, rmin, rmax, magmin, usable_out, rgb_out, nullptr);
}
int tfft_extract_batch(tfft_ctx* c, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins,
                       uint64_t n_bits, double alpha, uint8_t* bits_out) {
    if (!c || n_images < 0 || !rgb || !bins || !bits_out || n_bits == 0) return TFFT_E_INVALID;
    return batch_host(c, false, n_images, rgb, w, h, center, bins, nullptr, n_bits, alpha, 0, 0, 0, nullptr, nullptr, bits_out);
}
int tfft_embed_stream_batch(tfft_ctx* c, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins, uint64_t n_bins,
                            const uint8_t* header, const uint8_t* payload, uint64_t payload_len, double alpha, double rmin, double rmax,
                            double magmin, uint64_t* usable_out, uint8_t* rgb_out) {
    if (!c || n_images < 0 || !rgb || !rgb_out || !bins || !header || (payload_len && !payload) || n_bins == 0) return TFFT_E_INVALID;
    if (n_bins < 912 || payload_len > (n_bins - 912) / 56) return TFFT_E_INVALID;      // the walk is shorter than the stream (no wrap for huge lengths)
    StreamIO io; io.header_in = header; io.payload_in = payload; io.plen = payload_len;
    return batch_host(c, true, n_images, rgb, w, h, center, bins, nullptr, n_bins, alpha, rmin, rmax, magmin, usable_out, rgb_out, nullptr, &io);
}
int tfft_extract_stream_batch(tfft_ctx* c, int n_images, const uint8_t* rgb, int w, int h, int center, const tfft_bin* bins, uint64_t n_bins,
                              double alpha, uint8_t* header_out, uint8_t* payload_out, uint64_t max_payload_len, int32_t* status_out,
                              uint8_t* raw_bits_out) {
    if (!c || n_images < 0 || !rgb || !bins || n_bins == 0 || !header_out || !status_out || (max_payload_len && !payload_out)) return TFFT_E_INVALID;
    StreamIO io; io.header_out = header_out; io.payload_out = payload_out; io.max_plen = max_payload_len; io.status_out = status_out;
    return batch_host(c, false, n_images, rgb, w, h, center, bins, nullptr, n_bins, alpha, 0, 0, 0, nullptr, nullptr, raw_bits_out, &io);
}
void* tfft_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, 0) == hipSuccess ? p : nullptr;
}
void tfft_host_free(void* p) { if (p) (void)hipHostFree(p); }

int tfft_frame_expand_dev(tfft_ctx* c, int n_images, const void* header_dev, const void* payload_dev, uint64_t payload_len,
                          void* bits_out_dev) {
    if (!c || n_images < 0 || !header_dev || (payload_len && !payload_dev) || !bits_out_dev) return TFFT_E_INVALID;
    if (n_images == 0) return TFFT_OK;
    HIPCHK(c, launch_frame_expand((const uint8_t*)header_dev, (const uint8_t*)payload_dev, payload_len, n_images, (uint8_t*)bits_out_dev,
                                  38ull * 24 + payload_len * 56, c->stream));
    return TFFT_OK;
}
int tfft_frame_majority_dev(tfft_ctx* c, int n_images, const void* bits_dev, uint64_t payload_len, void* header_out_dev,
                            void* payload_out_dev) {
    if (!c || n_images < 0 || !bits_dev || !header_out_dev || (payload_len && !payload_out_dev)) return TFFT_E_INVALID;
    if (n_images == 0) return TFFT_OK;
    HIPCHK(c, launch_frame_majority((const uint8_t*)bits_dev, payload_len, n_images, (uint8_t*)header_out_dev, (uint8_t*)payload_out_dev, c->stream));
    return TFFT_OK;
}

int tfft_profile_stage(tfft_ctx* c, int n_images, int stage, int reps, const void* rgb_dev, void* rgb_out_dev,
                       const void* bins_dev, const void* bits_dev, void* bits_out_dev, uint64_t n_bits, double alpha,
                       float* ms_per_rep, int* n_launches) {
    if (!c || n_images < 1 || n_images > c->n_slots || reps < 1 || !ms_per_rep || stage < 0 || stage >= N_STAGES) return TFFT_E_INVALID;
    const Slot& s = c->slots[0];
    if (s.PH == 0) return TFFT_E_STATE;
    for (int i = 1; i < n_images; i++) c->slots[i] = s;
    const ColPlan pl = plan_cols(c, s.PH, s.PWi, n_images);
    int launches = 1;
    const int final_fwd = pl.direct ? COLS_FWD_A : COLS_FWD_B;
    if ((stage == COLS_FWD_B || stage == COLS_INV_B) && pl.direct) launches = 0;
    if ((stage == COLS_FWD_A || stage == COLS_INV_B) && pl.fused_fwd) launches = 0;
    if (stage == ROWS_FWD && pl.fused_fwd && c->fuse_live && s.H < s.PH && (s.H % (s.PH >> 3)) != 0) launches = 2;      // one per live-row count
    {
        const bool compact = c->stats_compact && (unsigned long long)s.PH * s.PWi <= (1ull << 24);
        const bool finish1 = compact && (unsigned long long)s.PH * s.PWi <= (1ull << 22) && n_images <= 4;
        if (stage == MEDIANS) launches = compact ? (c->median_force_fallback ? 2 : (finish1 ? 5 : 7)) + (c->stats_fused ? 1 : 0)
                                                 : (c->median_force_fallback ? 7 : 13) + (c->stats_fused ? 3 : 0);
    }
    if (stage == CAPACITY) launches = c->stats_fused ? 0 : 2;      // fused: counted inside the medians' full pass
    // delta embedding (see embed_chunk): the batched pipeline has no k_embed launch, its first inverse step builds the tiles from the
    // bins and its row kernel adds the cover -- the stages are timed the way the pipeline runs them
    bool delta = false;
    if (c->embed_delta && bins_dev && n_bits > 0 && (stage == EMBED || stage == COLS_INV_A || stage == ROWS_INV || stage == final_fwd || stage == MEDIANS)) {
        const EmbedParams ep0 = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
        delta = !ep0.generic;
    }
    CapParams tcap = cap_params(c, s, 0.05, 0.45);
    tcap.magmin = 0.01;
    // the statistics inside the last forward column step (see embed_chunk): that step is timed as COLS_STAT; MEDIANS is everything else
    // of the statistics -- sample pass + bracket guess before it, select chain, gated step, fallback and capacity kernels after it
    const bool tile = delta && bits_dev && tilestats_applies(c, s, pl, tcap, n_images) && (stage == final_fwd || stage == MEDIANS);
    const bool m2 = delta && bits_dev && !tile && stats_m2_applies(c, s, tcap);      // the spectrum is stored as |F|^2 + column 0 (see embed_chunk)
    if (tile && stage == MEDIANS) launches = 10;

    if (n_launches) *n_launches = launches;
    *ms_per_rep = 0.f;
    if (launches == 0) return TFFT_OK;
    { const float2* t; int rc = get_twiddles(c, s.PWi, &t); if (rc) return rc; rc = get_twiddles(c, s.PH, &t); if (rc) return rc; }
    ColParams rd{};
    if (stage == COLS_FWD_READ) {
        if (!bins_dev || !index_ok(c, n_bits)) return TFFT_E_INVALID;
        if (c->tile_read && n_bits > 0 && (n_images >= 8 || c->tile_read >= 2)) {
            if (!bits_out_dev) return TFFT_E_INVALID;
            const int G = pl.direct ? 1 : (1 << pl.log_n1), ntiles = (s.PWi / 2 + 15) / 16;
            int rc = ensure_buckets(c, 0, n_bits, 3 * ntiles * G);
            if (rc) return rc;
            HIPCHK(c, launch_bucket_bins((const tfft_bin*)bins_dev, c->bit_index, n_bits, s.PH, s.PWi, G, c->tb[0].cnt, c->tb[0].off, c->tb[0].ent, c->err, c->tile_read == 2, c->stream));
            rd.rd_bins = c->tb[0].ent; rd.rd_off = c->tb[0].off; rd.rd_bits = (uint8_t*)bits_out_dev; rd.rd_n = n_bits; rd.trash = c->trash;
        } else {
            HIPCHK(c, launch_bins_last_row((const tfft_bin*)bins_dev, n_bits, s.PH, s.PWi, c->last_row, c->stream));
        }
    }
    ColParams em{};
    if ((stage == COLS_INV_A || stage == final_fwd || stage == EMBED || (stage == MEDIANS && tile)) && delta && bits_dev) {
        if (!index_ok(c, n_bits)) return TFFT_E_STATE;
        const int G = pl.direct ? 1 : (1 << pl.log_n1), ntiles = (s.PWi / 2 + 15) / 16;
        int rc = ensure_buckets(c, 0, n_bits, 3 * ntiles * G, true);
        if (rc) return rc;
        rc = build_buckets(c, 0, (const tfft_bin*)bins_dev, n_bits, s, G, c->stream);
        if (rc) return rc;
        const EmbedParams ep0 = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
        em.rd_bins = c->tb[0].ent; em.rd_off = c->tb[0].off; em.trash = c->trash; em.em_fl = c->tb[0].fl; em.em_pb = c->tb[0].pb; em.em_n = n_bits;
        em.em_cos = ep0.cos_a; em.em_sin = ep0.sin_a;
        if (m2) { em.em_m2 = 1; em.st_col0 = c->col0_pool; }
    }
    if (tile) {
        // phases of enqueue_forward_tilestats: 2 sample + guess, 4 COLS_STAT, 8 select, 16 tail.  The COLS_STAT step alone needs a bracket
        // (one untimed sample pass); MEDIANS = the whole sequence minus the COLS_STAT launches it contains, timed the same way
        float ms_all = 0.f, ms_stat = 0.f;
        int rc = enqueue_forward_tilestats(c, 0, n_images, (const uint8_t*)rgb_dev, c->stream, em, tcap, c->usable, 2);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
        for (int r = 0; r < reps; r++) { rc = enqueue_forward_tilestats(c, 0, n_images, (const uint8_t*)rgb_dev, c->stream, em, tcap, c->usable, 4); if (rc) return rc; }
        HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev_t1));
        HIPCHK(c, hipEventElapsedTime(&ms_stat, c->ev_t0, c->ev_t1));
        if (stage == MEDIANS) {
            HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
            for (int r = 0; r < reps; r++) { rc = enqueue_forward_tilestats(c, 0, n_images, (const uint8_t*)rgb_dev, c->stream, em, tcap, c->usable, 2 | 4 | 8 | 16); if (rc) return rc; }
            HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev_t1));
            HIPCHK(c, hipEventElapsedTime(&ms_all, c->ev_t0, c->ev_t1));
        }
        *ms_per_rep = (stage == MEDIANS ? (ms_all > ms_stat ? ms_all - ms_stat : 0.f) : ms_stat) / (float)reps;
        return TFFT_OK;
    }
    HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
    for (int r = 0; r < reps; r++) {
        int rc = TFFT_OK;
        switch (stage) {
            case COLS_FWD_A:
            case COLS_FWD_B:
                { StageMode md; if (em.rd_bins && stage == final_fwd) md.fwd_emit = &em;
                  rc = enqueue_fft_stage(c, 0, n_images, stage, (const uint8_t*)rgb_dev, nullptr, c->stream, md); }
                break;
            case COLS_INV_A:
                { StageMode md; if (em.rd_bins) md.inv_embed = &em;
                  rc = enqueue_fft_stage(c, 0, n_images, stage, nullptr, nullptr, c->stream, md); }
                break;
            case ROWS_INV:
                { StageMode md; if (delta && rgb_dev) md.inv_cover = (const uint8_t*)rgb_dev;
                  rc = enqueue_fft_stage(c, 0, n_images, stage, nullptr, (uint8_t*)rgb_out_dev, c->stream, md); }
                break;
            case COLS_FWD_READ:
                { StageMode md; if (rd.rd_bins) md.fwd_read = &rd; else md.fwd_last_row = c->last_row;
                  rc = enqueue_fft_stage(c, 0, n_images, final_fwd, nullptr, nullptr, c->stream, md); }
                break;
            case EMBED: {
                if (!index_ok(c, n_bits)) return TFFT_E_STATE;
                if (em.rd_bins) {      // delta embedding: what is left of the embed stage is the gather of the stream bits into bucket order
                    const int G = pl.direct ? 1 : (1 << pl.log_n1), nb = 3 * ((s.PWi / 2 + 15) / 16) * G;
                    HIPCHK(c, launch_gather_bits(c->tb[0].ent, c->tb[0].off + nb, (const uint8_t*)bits_dev, nullptr, nullptr, 0, n_bits, n_bits, n_images, c->tb[0].pb, c->stream));
                    break;
                }
                EmbedParams ep = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
                HIPCHK(c, launch_embed(c->spec(0), (const tfft_bin*)bins_dev, (const uint8_t*)bits_dev, nullptr, ep, n_images, c->err, c->stream));
                break;
            }
            case READ: {
                if (!index_ok(c, n_bits)) return TFFT_E_STATE;
                EmbedParams ep = embed_params(c, s, n_bits, alpha, 0, nullptr, false);
                HIPCHK(c, launch_read(c->spec(0), (const tfft_bin*)bins_dev, nullptr, ep, n_images, (uint8_t*)bits_out_dev, c->err, c->stream));
                break;
            }
            case MEDIANS:
                if (c->stats_fused) {
                    CapParams p = cap_params(c, s, 0.05, 0.45);
                    p.magmin = 0.01;
                    rc = enqueue_medians(c, 0, n_images, c->stream, &p, c->usable, m2);
                } else rc = enqueue_medians(c, 0, n_images, c->stream);
                break;
            case CAPACITY: {
                CapParams p = cap_params(c, s, 0.05, 0.45);
                p.magmin = 0.01;
                HIPCHK(c, launch_capacity(c->spec(0), p, n_images, c->med, c->partial, c->usable, c->stream));
                break;
            }
            default: rc = enqueue_fft_stage(c, 0, n_images, stage, (const uint8_t*)rgb_dev, (uint8_t*)rgb_out_dev, c->stream);
        }
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev_t1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
    *ms_per_rep = ms / (float)reps;
    return TFFT_OK;
}

// ---------------------------------------------------------------- (f-4) fp64 audit transform
static int audit_run(tfft_ctx* c, double* host, const uint8_t* rgb, int w, int h, int center, int n_planes, int ph, int pw, int inverse) {
    const size_t bytes = (size_t)n_planes * ph * pw * sizeof(double2);
    double2 *a = nullptr, *scratch = nullptr, *wtab = nullptr;
    uint8_t* img = nullptr;
    int rc = TFFT_OK;
    auto done = [&](int r) { (void)hipFree(a); (void)hipFree(scratch); (void)hipFree(wtab); (void)hipFree(img); return r; };
    if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&scratch, bytes) != hipSuccess ||
        hipMalloc((void**)&wtab, (size_t)(ph > pw ? ph : pw) * sizeof(double2)) != hipSuccess) return done(TFFT_E_NOMEM);
    hipError_t e;
    if (rgb) {
        const size_t ib = (size_t)w * h * 3;
        if (hipMalloc((void**)&img, ib) != hipSuccess) return done(TFFT_E_NOMEM);
        e = hipMemcpyAsync(img, rgb, ib, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = audit_load_rgb8_f64(img, w, h, pw, ph, center, a, c->stream);
    } else {
        e = hipMemcpyAsync(a, host, bytes, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = audit_fft2d_f64(a, scratch, wtab, n_planes, ph, pw, inverse, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host, a, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->last_hip = (int)e; rc = TFFT_E_HIP; }
    return done(rc);
}
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int tfft_audit_fft2d_f64(tfft_ctx* c, double* planes, int n_planes, int ph, int pw, int inverse) {
    if (!c || !planes || n_planes < 1 || !is_pow2(ph) || !is_pow2(pw) || ph > TFFT_MAX_DIM || pw > TFFT_MAX_DIM) return TFFT_E_INVALID;
    return audit_run(c, planes, nullptr, 0, 0, 0, n_planes, ph, pw, inverse);
}

int tfft_audit_forward_rgb8_f64(tfft_ctx* c, const uint8_t* rgb, int w, int h, int center, double* out) {
    if (!c || !rgb || !out || w < 1 || h < 1 || w > TFFT_MAX_DIM || h > TFFT_MAX_DIM) return TFFT_E_INVALID;
    int pw = 1, ph = 1;
    while (pw < w) pw <<= 1;
    while (ph < h) ph <<= 1;
    return audit_run(c, out, rgb, w, h, center, 3, ph, pw, 0);
}

int tfft_timer_begin(tfft_ctx* c) {
    if (!c) return TFFT_E_INVALID;
    HIPCHK(c, hipEventRecord(c->ev_t0, c->stream));
    return TFFT_OK;
}
int tfft_timer_end(tfft_ctx* c, float* ms) {
    if (!c || !ms) return TFFT_E_INVALID;
    HIPCHK(c, hipEventRecord(c->ev_t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev_t1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev_t0, c->ev_t1));
    return TFFT_OK;
}

}  // extern "C"
