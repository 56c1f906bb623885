// tfft_audit64.hip --  the reference's fft2d evaluated in fp64 on the device.
//
// Not a fast path: an AUDIT transform.  It follows steganosaur.cpp:341-366 operation by operation --
// bit reversal (S:343-345), radix-2 DIT stages with exp(+-2*pi*i/len) (S:346-356), the stage twiddles
// produced by the same recurrence w *= wlen starting from 1 (S:353), plain multiply/add without FMA
// contraction (this file is compiled with -ffp-contract=off), rows first, then columns (S:359-366),
// inverse divided by n per dimension (S:357) -- so its output equals the CPU reference bit for bit
// (tests/test_emulated.py and tests/test_gpu_parity.py check that against the oracle).  What it is for:
// measuring the fp32 product path against the reference's arithmetic at sizes where the CPU reference
// takes minutes (4096^2, 8192^2), on the GPU box where the reference does not exist.
//
// The recurrence makes w_j depend on (len, j) only, so the host computes one table per dimension with
// exactly the reference's host arithmetic (cos/sin of the same libm, the same sequence of complex
// products) and the stages read it: stage len uses entries [len/2 - 1, len - 1).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <vector>

#include "tfft_kernels.h"

namespace tfft {

struct Seq64 {           // a batch of strided sequences inside n_planes planes
    int n, logn;         // length (power of two)
    long long nseq;      // sequences per plane
    long long seq_stride, elem_stride, plane_stride;     // in double2 elements
    int n_planes;
};

// out[rev(i)] = in[i] for every sequence (S:343-345 is this permutation done by swaps)
__global__ void k64_bitrev(const double2* __restrict__ in, double2* __restrict__ out, Seq64 q) {
    const long long per_plane = q.nseq * q.n, total = per_plane * q.n_planes;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / per_plane, r = e - p * per_plane;
        const long long s = r / q.n;
        const unsigned i = (unsigned)(r - s * q.n);
        unsigned j = 0;
        for (int b = 0; b < q.logn; b++) j |= ((i >> b) & 1u) << (q.logn - 1 - b);
        const long long base = p * q.plane_stride + s * q.seq_stride;
        out[base + (long long)j * q.elem_stride] = in[base + (long long)i * q.elem_stride];
    }
}

// one radix-2 stage (S:349-355): u = a[i+j], v = a[i+j+len/2]*w_j; a[i+j] = u+v; a[i+j+len/2] = u-v
__global__ void k64_stage(double2* __restrict__ a, const double2* __restrict__ wtab, int len, Seq64 q) {
    const int half = len >> 1;
    const long long per_seq = q.n >> 1, per_plane = q.nseq * per_seq, total = per_plane * q.n_planes;
    const double2* w = wtab + (half - 1);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long p = e / per_plane, r = e - p * per_plane;
        const long long s = r / per_seq;
        const int b = (int)(r - s * per_seq);
        const int j = b & (half - 1), i = (b - j) * 2;
        double2* pu = a + p * q.plane_stride + s * q.seq_stride + (long long)(i + j) * q.elem_stride;
        double2* pv = pu + (long long)half * q.elem_stride;
        const double2 u = *pu, x = *pv, wj = w[j];
        const double vr = x.x * wj.x - x.y * wj.y, vi = x.x * wj.y + x.y * wj.x;
        *pu = make_double2(u.x + vr, u.y + vi);
        *pv = make_double2(u.x - vr, u.y - vi);
    }
}

// S:357: z /= double(n)
__global__ void k64_divide(double2* __restrict__ a, double n, long long total) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        double2 z = a[e];
        a[e] = make_double2(z.x / n, z.y / n);
    }
}

// to_planes_u8 / apply_center / pad_to_fft (S:383-398) into fp64 planes [3][PH][PW]
__global__ void k64_load_rgb8(const uint8_t* __restrict__ rgb, int W, int H, int PW, int PH, int center, double2* __restrict__ out) {
    const long long P = (long long)PW * PH, total = 3 * P;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(e / P);
        const long long r = e - (long long)p * P;
        const int y = (int)(r / PW), x = (int)(r - (long long)y * PW);
        double v = 0.0;
        if (x < W && y < H) {
            v = (double)rgb[((size_t)y * W + x) * 3 + p];
            if (center && ((x + y) & 1)) v = -v;
        }
        out[e] = make_double2(v, 0.0);
    }
}

// stage twiddles of one dimension by the reference's recurrence (host arithmetic, S:347-353)
static void stage_twiddles(int n, bool inverse, std::vector<double2>& tab) {
    tab.assign(n > 1 ? n - 1 : 1, make_double2(1.0, 0.0));
    for (size_t len = 2; len <= (size_t)n; len <<= 1) {
        const double ang = 2 * M_PI / len * (inverse ? -1 : 1);
        const double lr = cos(ang), li = sin(ang);
        double wr = 1, wi = 0;
        for (size_t j = 0; j < len / 2; j++) {
            tab[len / 2 - 1 + j] = make_double2(wr, wi);
            const double nr = wr * lr - wi * li, ni = wr * li + wi * lr;
            wr = nr; wi = ni;
        }
    }
}

static hipError_t fft_dim(double2* a, double2* scratch, const Seq64& q, bool inverse, double2* wtab_dev, hipStream_t s) {
    if (q.n < 2) return hipSuccess;
    std::vector<double2> tab;
    stage_twiddles(q.n, inverse, tab);
    hipError_t e = hipMemcpyAsync(wtab_dev, tab.data(), tab.size() * sizeof(double2), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);      // `tab` is a local
    if (e != hipSuccess) return e;
    const long long total = q.nseq * q.n * q.n_planes;
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 65535u * 4) nb = 65535u * 4;
    hipLaunchKernelGGL(k64_bitrev, dim3(nb), dim3(256), 0, s, a, scratch, q);
    for (int len = 2; len <= q.n; len <<= 1) {
        unsigned nbs = (unsigned)((total / 2 + 255) / 256);
        if (nbs > 65535u * 4) nbs = 65535u * 4;
        if (nbs < 1) nbs = 1;
        hipLaunchKernelGGL(k64_stage, dim3(nbs), dim3(256), 0, s, scratch, wtab_dev, len, q);
    }
    return hipGetLastError();
}

// In-place fp64 2-D FFT of n_planes planes [PH][PW] (device pointer `a`, scratch of the same size, wtab of
// max(PH,PW) entries).  The bit-reversal pass writes scratch, the stages work on scratch, and the result of
// each dimension is copied back by the NEXT dimension's bit reversal (or the final copy).
hipError_t audit_fft2d_f64(double2* a, double2* scratch, double2* wtab, int n_planes, int PH, int PW, int inverse, hipStream_t s) {
    int lw = 0, lh = 0;
    while ((1 << lw) < PW) lw++;
    while ((1 << lh) < PH) lh++;
    if ((1 << lw) != PW || (1 << lh) != PH) return hipErrorInvalidValue;
    const long long P = (long long)PW * PH;
    const size_t bytes = (size_t)P * n_planes * sizeof(double2);
    // rows (S:361): a -> scratch
    Seq64 rows{PW, lw, PH, PW, 1, P, n_planes};
    hipError_t e;
    if (PW >= 2) { e = fft_dim(a, scratch, rows, inverse != 0, wtab, s); if (e != hipSuccess) return e; }
    else { e = hipMemcpyAsync(scratch, a, bytes, hipMemcpyDeviceToDevice, s); if (e != hipSuccess) return e; }
    if (inverse && PW >= 2) {
        const long long total = P * n_planes;
        hipLaunchKernelGGL(k64_divide, dim3((unsigned)((total + 255) / 256 > 262140 ? 262140 : (total + 255) / 256)), dim3(256), 0, s, scratch, (double)PW, total);
    }
    // columns (S:362-365): scratch -> a
    Seq64 cols{PH, lh, PW, 1, PW, P, n_planes};
    if (PH >= 2) { e = fft_dim(scratch, a, cols, inverse != 0, wtab, s); if (e != hipSuccess) return e; }
    else { e = hipMemcpyAsync(a, scratch, bytes, hipMemcpyDeviceToDevice, s); if (e != hipSuccess) return e; }
    if (inverse && PH >= 2) {
        const long long total = P * n_planes;
        hipLaunchKernelGGL(k64_divide, dim3((unsigned)((total + 255) / 256 > 262140 ? 262140 : (total + 255) / 256)), dim3(256), 0, s, a, (double)PH, total);
    }
    return hipGetLastError();
}

hipError_t audit_load_rgb8_f64(const uint8_t* rgb_dev, int W, int H, int PW, int PH, int center, double2* out, hipStream_t s) {
    const long long total = 3LL * PW * PH;
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 262140u) nb = 262140u;
    hipLaunchKernelGGL(k64_load_rgb8, dim3(nb), dim3(256), 0, s, rgb_dev, W, H, PW, PH, center, out);
    return hipGetLastError();
}

}  // namespace tfft
