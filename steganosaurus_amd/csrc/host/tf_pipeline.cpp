// tf_pipeline.cpp -- SURVEY 8 f-1 end to end: PNG files -> GPU -> PNG files for a batch of equal-sized covers, the three stages
// overlapped (replaces the reference's stbi_load S:909 / stbi_write_png S:1104 around do_embed / do_extract for batches):
//   worker threads inflate the PNGs of chunk k+1 into a pinned buffer      (tf_png.cpp, zlib)
//   the calling thread runs chunk k through tfft_embed_stream_batch        (H2D, kernels and D2H overlapped inside, tfft_capi.hip)
//   worker threads deflate and write the stego images of chunk k-1
// Built as steganosaurus_amd/libtfpipe.so; plain C ABI (include/turtlefft_pipe.h).  The device context stays single-threaded: only
// the calling thread touches it.
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/turtlefft_hip.h"
#include "../../../include/turtlefft_pipe.h"
#include "tf_png.h"

namespace {

class Pool {      // fixed worker threads + a FIFO of tasks; wait() = all tasks submitted so far are done
public:
    explicit Pool(int n) {
        for (int i = 0; i < (n < 1 ? 1 : n); i++) th_.emplace_back([this] { run(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> l(m_); q_.push(std::move(f)); pending_++; }
        cv_.notify_one();
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
private:
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                f = std::move(q_.front()); q_.pop();
            }
            f();
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::queue<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    int pending_ = 0;
    bool stop_ = false;
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Chunk { uint8_t* in = nullptr; uint8_t* out = nullptr; int first = 0, n = 0; };

}  // namespace

extern "C" int tfp_embed_png_batch(tfft_ctx* ctx, int n_files, const char* const* in_paths, const char* const* out_paths, int w, int h, int center,
                                   const tfft_bin* bins, uint64_t n_bins, const uint8_t* headers, const uint8_t* payloads, uint64_t payload_len,
                                   double alpha, double rmin, double rmax, double magmin, int chunk_images, int n_threads, int png_level,
                                   uint64_t* usable_out, double* stage_ms) {
    if (!ctx || n_files < 0 || !in_paths || !out_paths || w < 1 || h < 1 || chunk_images < 1 || (n_bins && !bins) || !headers || (payload_len && !payloads))
        return TFFT_E_INVALID;
    const size_t img = (size_t)w * h * 3;
    Chunk ring[3];
    for (auto& c : ring) {
        c.in = (uint8_t*)tfft_host_alloc(img * chunk_images);
        c.out = (uint8_t*)tfft_host_alloc(img * chunk_images);
        if (!c.in || !c.out) { for (auto& d : ring) { tfft_host_free(d.in); tfft_host_free(d.out); } return TFFT_E_NOMEM; }
    }
    std::atomic<int> fail{0};
    double t_dec = 0, t_gpu = 0, t_enc = 0;
    std::mutex tm;
    const double t_all0 = now_ms();
    {
        Pool dec(n_threads), enc(n_threads);
        const int n_chunks = (n_files + chunk_images - 1) / chunk_images;
        auto start_decode = [&](int k) {
            Chunk& c = ring[k % 3];
            c.first = k * chunk_images;
            c.n = n_files - c.first < chunk_images ? n_files - c.first : chunk_images;
            for (int i = 0; i < c.n; i++) {
                const int f = c.first + i;
                uint8_t* dst = c.in + (size_t)i * img;
                dec.submit([&, f, dst] {
                    const double t0 = now_ms();
                    std::vector<uint8_t> rgb; int iw = 0, ih = 0;
                    if (!tfh::load_rgb8(in_paths[f], rgb, iw, ih) || iw != w || ih != h) { fail = 1; return; }
                    memcpy(dst, rgb.data(), img);
                    const double dt = now_ms() - t0;
                    std::lock_guard<std::mutex> l(tm); t_dec += dt;
                });
            }
        };
        if (n_chunks > 0) start_decode(0);
        int rc = TFFT_OK;
        for (int k = 0; k < n_chunks && rc == TFFT_OK && !fail; k++) {
            dec.wait();                                   // chunk k is in its pinned buffer
            if (fail) break;
            if (k + 1 < n_chunks) {
                if (k >= 2) enc.wait();                   // ring slot (k+1) % 3 was chunk k-2: its encoders must be done with it
                start_decode(k + 1);
            }
            Chunk& c = ring[k % 3];
            const double t0 = now_ms();
            rc = tfft_embed_stream_batch(ctx, c.n, c.in, w, h, center, bins, n_bins, headers + (size_t)c.first * 38,
                                         payloads ? payloads + (size_t)c.first * payload_len : nullptr, payload_len, alpha, rmin, rmax, magmin,
                                         usable_out ? usable_out + c.first : nullptr, c.out);
            t_gpu += now_ms() - t0;
            if (rc != TFFT_OK) break;
            for (int i = 0; i < c.n; i++) {
                const int f = c.first + i;
                const uint8_t* src = c.out + (size_t)i * img;
                enc.submit([&, f, src] {
                    const double t1 = now_ms();
                    std::vector<uint8_t> png;
                    bool ok = tfh::png_encode_rgb8_opt(src, w, h, png, png_level > 0 ? png_level : 6, png_level <= 0 || png_level >= 6);
                    if (ok) {
                        FILE* fp = fopen(out_paths[f], "wb");
                        ok = fp && fwrite(png.data(), 1, png.size(), fp) == png.size();
                        if (fp) ok = (fclose(fp) == 0) && ok;
                    }
                    if (!ok) fail = 2;
                    const double dt = now_ms() - t1;
                    std::lock_guard<std::mutex> l(tm); t_enc += dt;
                });
            }
        }
        dec.wait(); enc.wait();
        if (rc != TFFT_OK) { for (auto& d : ring) { tfft_host_free(d.in); tfft_host_free(d.out); } return rc; }
    }
    for (auto& d : ring) { tfft_host_free(d.in); tfft_host_free(d.out); }
    if (stage_ms) { stage_ms[0] = now_ms() - t_all0; stage_ms[1] = t_dec; stage_ms[2] = t_gpu; stage_ms[3] = t_enc; }
    if (fail == 1) return TFFT_E_INVALID;      // a file that is not a w x h image
    if (fail == 2) return TFFT_E_STATE;        // PNG encode / write failed
    return TFFT_OK;
}

extern "C" int tfp_extract_png_batch(tfft_ctx* ctx, int n_files, const char* const* in_paths, int w, int h, int center, const tfft_bin* bins,
                                     uint64_t n_bins, double alpha, uint8_t* headers_out, uint8_t* payloads_out, uint64_t max_payload_len,
                                     int32_t* status_out, int chunk_images, int n_threads, double* stage_ms) {
    if (!ctx || n_files < 0 || !in_paths || w < 1 || h < 1 || chunk_images < 1 || !bins || !headers_out || !status_out) return TFFT_E_INVALID;
    const size_t img = (size_t)w * h * 3;
    uint8_t* ring[2] = {(uint8_t*)tfft_host_alloc(img * chunk_images), (uint8_t*)tfft_host_alloc(img * chunk_images)};
    if (!ring[0] || !ring[1]) { tfft_host_free(ring[0]); tfft_host_free(ring[1]); return TFFT_E_NOMEM; }
    std::atomic<int> fail{0};
    double t_dec = 0, t_gpu = 0;
    std::mutex tm;
    const double t_all0 = now_ms();
    int rc = TFFT_OK;
    {
        Pool dec(n_threads);
        const int n_chunks = (n_files + chunk_images - 1) / chunk_images;
        auto start_decode = [&](int k) {
            const int first = k * chunk_images, n = n_files - first < chunk_images ? n_files - first : chunk_images;
            for (int i = 0; i < n; i++) {
                const int f = first + i;
                uint8_t* dst = ring[k & 1] + (size_t)i * img;
                dec.submit([&, f, dst] {
                    const double t0 = now_ms();
                    std::vector<uint8_t> rgb; int iw = 0, ih = 0;
                    if (!tfh::load_rgb8(in_paths[f], rgb, iw, ih) || iw != w || ih != h) { fail = 1; return; }
                    memcpy(dst, rgb.data(), img);
                    const double dt = now_ms() - t0;
                    std::lock_guard<std::mutex> l(tm); t_dec += dt;
                });
            }
        };
        if (n_chunks > 0) start_decode(0);
        for (int k = 0; k < n_chunks && rc == TFFT_OK && !fail; k++) {
            dec.wait();
            if (fail) break;
            const int first = k * chunk_images, n = n_files - first < chunk_images ? n_files - first : chunk_images;
            // (the extraction call returns only when its D2H copies are done, so the other ring slot is free for the next decode now)
            if (k + 1 < n_chunks) start_decode(k + 1);
            const double t0 = now_ms();
            rc = tfft_extract_stream_batch(ctx, n, ring[k & 1], w, h, center, bins, n_bins, alpha, headers_out + (size_t)first * 38,
                                           payloads_out ? payloads_out + (size_t)first * max_payload_len : nullptr, max_payload_len, status_out + first, nullptr);
            t_gpu += now_ms() - t0;
        }
        dec.wait();
    }
    tfft_host_free(ring[0]); tfft_host_free(ring[1]);
    if (stage_ms) { stage_ms[0] = now_ms() - t_all0; stage_ms[1] = t_dec; stage_ms[2] = t_gpu; stage_ms[3] = 0; }
    if (rc != TFFT_OK) return rc;
    return fail ? TFFT_E_INVALID : TFFT_OK;
}
