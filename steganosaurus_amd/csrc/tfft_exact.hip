// tfft_exact.hip -- the two integer-valued outputs of the path that an fp32 spectrum cannot reproduce on its own, made exact:
// median_abs (S:404-409, the element at sorted index P/2 of |F|) and count_plane (S:998-1008, bins with |F| >= thr).
//
// The fp32 transform is within ~1e-7 * rms of the reference's fp64 one, so its order statistic picks the right NEIGHBOURHOOD of
// the sorted sequence but not necessarily the right element, and a bin within rounding of the threshold can fall on the wrong
// side.  Both decisions only ever concern a handful of bins: those whose fp32 magnitude lies within a small window around the
// fp32 median / the threshold.  k_exact_collect finds them (and counts everything safely below / above the window),
// k_exact_eval recomputes each of them in fp64 straight from the pixels -- a direct 2-D DFT sum of the u8 image at that one
// frequency, with twiddles looked up by exact integer index in a table of correctly rounded values, so every term carries
// one rounding -- and the host settles rank and count on those values (tfft_capi.hip).  The result equals the reference's up to
// the ~1e-14 by which two fp64 summation orders differ: medians to ~1e-13 relative, counts exactly unless a bin sits within
// that distance of the threshold.
//
// Cost: one more pass over the stored spectrum + a few dozen 256-thread workgroups per candidate walking the image (L2 resident):
// a fraction of a millisecond for a 1080p image, a few host round trips.  That is why it serves the single-image calls (tfft_medians / tfft_capacity: what the CLI uses
// for "Message too large") and not the batched pipelines, whose statistics stay on the fp32 planes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "tfft_kernels.h"

namespace tfft {

extern __shared__ __attribute__((aligned(16))) unsigned char tfft_smem[];

// F[y][0] and F[y][M] out of the packed column 0 (as in tfft_kernels.hip)
__device__ __forceinline__ void exact_unpack_col0(const float2* __restrict__ plane, int y, int PH, int M, float2& f0, float2& fm) {
    const float2 a = plane[(size_t)y * M], b = plane[(size_t)((PH - y) & (PH - 1)) * M];
    f0 = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    fm = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));
}

// One pass over the half spectrum of one image (3 planes, grid.y = plane).  A stored bin (y, x), 0 < x < M, stands for the
// full-grid bins (y, x) and ((PH-y)%PH, PW-x), which share their magnitude; the packed column 0 holds (y, 0) and (y, M), one each.
//   median mode (P.cap == 0): weight of everything below the window -> below[plane]; bins inside it -> candidates (weight 2 / 1)
//   capacity mode: bins of the annulus, off the axes (S:698-700, S:1003) -- the stored bin and its mirror tested separately --
//                  above the window -> above[plane] (weighted); inside -> candidates with that weight
__global__ void k_exact_collect(const float2* __restrict__ spec, ExactCollect P, ExactCand* __restrict__ cand, unsigned long long* __restrict__ below,
                                unsigned* __restrict__ n_cand) {
    const int plane = blockIdx.y, M = P.PW >> 1;
    const float2* pl = spec + (size_t)plane * P.PH * M;
    const float lo2 = P.lo2[plane], hi2 = P.hi2[plane];
    ExactCand* out = cand + (size_t)plane * P.cap_cand;
    unsigned long long acc = 0;
    auto emit = [&](int y, int x, unsigned w, float m2) {
        const unsigned slot = atomicAdd(&n_cand[plane], 1u);
        if (slot < (unsigned)P.cap_cand) { ExactCand e; e.y = (uint16_t)y; e.x = (uint16_t)x; e.w = (uint16_t)w; e.plane = (uint16_t)plane; e.m2 = m2; out[slot] = e; }
    };
    for (int y = blockIdx.x; y < P.PH; y += gridDim.x) {
        for (int x = threadIdx.x; x < M; x += blockDim.x) {
            if (x == 0) {
                if (P.cap) continue;                          // columns 0 and PW/2 are excluded axes
                float2 f0, fm;
                exact_unpack_col0(pl, y, P.PH, M, f0, fm);
                const float a = fmaf(f0.x, f0.x, f0.y * f0.y), b = fmaf(fm.x, fm.x, fm.y * fm.y);
                if (a < lo2) acc += 1; else if (a <= hi2) emit(y, 0, 1u, a);
                if (b < lo2) acc += 1; else if (b <= hi2) emit(y, M, 1u, b);
                continue;
            }
            const float2 v = pl[(size_t)y * M + x];
            const float m2 = fmaf(v.x, v.x, v.y * v.y);
            if (!P.cap) {
                if (m2 < lo2) acc += 2; else if (m2 <= hi2) emit(y, x, 2u, m2);
            } else {
                unsigned w = 0;
                if (y != 0 && 2 * y != P.PH) {
                    const unsigned long long d1 = (unsigned long long)y * y + (unsigned long long)x * x;
                    w = (d1 >= P.s_lo && d1 <= P.s_hi) ? 1u : 0u;
                    const unsigned long long ym = (unsigned long long)(P.PH - y), xm = (unsigned long long)(P.PW_full - x);
                    const unsigned long long d2 = ym * ym + xm * xm;
                    w += (d2 >= P.s_lo && d2 <= P.s_hi) ? 1u : 0u;
                }
                if (!w) continue;
                if (m2 > hi2) acc += w; else if (m2 >= lo2) emit(y, x, w, m2);
            }
        }
    }
    // one 64-bit atomic per block
    unsigned long long* red = reinterpret_cast<unsigned long long*>(tfft_smem);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0 && red[0]) atomicAdd(&below[plane], red[0]);
}

// T[j] = exp(2 pi i j/PW), j < PW, correctly rounded (sincospi): built once per row length, kept by the context
__global__ void k_exact_table(double2* __restrict__ T, int PW) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= PW) return;
    double s, c;
    sincospi(2.0 * (double)j / (double)PW, &s, &c);
    T[j] = make_double2(c, s);
}

// F[y][x] of one colour plane of the u8 image in fp64: sum over the W x H pixels of s(r,c) * pix * exp(+2 pi i (y r/PH + x c/PW))
// (the reference's sign convention, S:347; s = (-1)^(r+c) with --center, folded into the frequencies: x + PW/2, y + PH/2).
// A candidate's rows are dealt to gridDim.y workgroups of 256 threads (16 threads walk a row: pixel columns cc, cc+16, ...; 16 rows
// per sweep); each writes its partial sum, the host adds them in a fixed order.  The column twiddles come out of an LDS copy of the
// table by the exact integer index (x*c) mod PW, so every term of the sum carries one rounding; the row factor is one sincospi per row.
//   grid (n_cand, n_split)
__global__ void __launch_bounds__(256) k_exact_eval(const uint8_t* __restrict__ rgb, int W, int H, int PW, int PH, int center,
                                                     const ExactCand* __restrict__ cand, const double2* __restrict__ table, double2* __restrict__ out) {
    double2* T = reinterpret_cast<double2*>(tfft_smem);
    const ExactCand cd = cand[blockIdx.x];
    const int tid = threadIdx.x, rr = tid >> 4, cc = tid & 15;
    for (int j = tid; j < PW; j += 256) T[j] = table[j];
    __syncthreads();
    const unsigned xe = ((unsigned)cd.x + (center ? (unsigned)(PW >> 1) : 0u)) & (unsigned)(PW - 1);
    const unsigned ye = ((unsigned)cd.y + (center ? (unsigned)(PH >> 1) : 0u)) & (unsigned)(PH - 1);
    const unsigned pmask = (unsigned)(PW - 1);
    const int plane = cd.plane;
    const int rows_per = (H + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = (r0 + rows_per < H) ? r0 + rows_per : H;
    double tre = 0.0, tim = 0.0;
    for (int r = r0 + rr; r < r1; r += 16) {
        const uint8_t* row = rgb + (size_t)r * W * 3 + plane;
        double are = 0.0, aim = 0.0;
        unsigned idx = (xe * (unsigned)cc) & pmask;
        const unsigned step = (xe * 16u) & pmask;
        for (int c = cc; c < W; c += 16) {
            const double p = (double)row[3 * c];
            const double2 w = T[idx];
            are = fma(p, w.x, are); aim = fma(p, w.y, aim);
            idx = (idx + step) & pmask;
        }
        double s, c;
        sincospi(2.0 * (double)((ye * (unsigned)r) & (unsigned)(PH - 1)) / (double)PH, &s, &c);
        tre += are * c - aim * s;
        tim += are * s + aim * c;
    }
    __syncthreads();                    // the table is dead: its space carries the reduction
    double2* red = reinterpret_cast<double2*>(tfft_smem);
    red[tid] = make_double2(tre, tim);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[tid].x += red[tid + s].x; red[tid].y += red[tid + s].y; }
        __syncthreads();
    }
    if (tid == 0) out[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = red[0];
}

hipError_t launch_exact_collect(const float2* spec, const ExactCollect& P, ExactCand* cand, unsigned long long* below, unsigned* n_cand, hipStream_t s) {
    hipError_t e = hipMemsetAsync(below, 0, 3 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(n_cand, 0, 3 * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    int nb = P.PH < 1024 ? P.PH : 1024;
    hipLaunchKernelGGL(k_exact_collect, dim3(nb, 3), dim3(256), 256 * sizeof(unsigned long long), s, spec, P, cand, below, n_cand);
    return hipGetLastError();
}

hipError_t launch_exact_table(double2* table, int PW, hipStream_t s) {
    hipLaunchKernelGGL(k_exact_table, dim3((PW + 255) / 256), dim3(256), 0, s, table, PW);
    return hipGetLastError();
}

// out: n * n_split partial sums, candidate-major
hipError_t launch_exact_eval(const uint8_t* rgb, int W, int H, int PW, int PH, int center, const ExactCand* cand, unsigned n, int n_split,
                             const double2* table, double2* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t lds = (size_t)(PW > 256 ? PW : 256) * sizeof(double2);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_exact_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_exact_eval, dim3(n, n_split), dim3(256), lds, s, rgb, W, H, PW, PH, center, cand, table, out);
    return hipGetLastError();
}

}  // namespace tfft
