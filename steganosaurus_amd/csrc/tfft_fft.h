// tfft_fft.h -- register/LDS Stockham FFT building blocks for gfx950 (wave64).
//
// A length-N complex FFT is done by N/E threads, each holding E elements in
// VGPRs.  Thread t always owns the elements at positions t + m*(N/E), m < E --
// both on entry (time index) and on exit (frequency index, natural order) --
// so the first pass can be fed straight from global memory and the last pass
// can be stored straight back: LDS is touched only for the exchanges between
// radix passes (Stockham autosort, no bit-reversal pass).
//
// Sign convention is the reference's (steganosaur.cpp:347): SIGN=+1 is the
// "forward" kernel exp(+2*pi*i*nk/N), SIGN=-1 the inverse kernel.  No scaling
// is applied here.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace tfft {

// Complex arithmetic on native 2-vectors: (re, im) stays ONE 64-bit register pair through the optimiser, so an add is one
// v_pk_add_f32, a multiply by a complex constant one v_pk_mul_f32 + one v_pk_fma_f32, and a multiplication by +-i only a source
// swizzle (op_sel / neg) of whatever consumes it.  Written component by component (make_float2(a.x + b.x, ...)) the vectoriser
// paired components of DIFFERENT values and patched the pairs back together with v_mov_b32: 22 % of the VALU instructions of the
// fused row+column kernels, which are VALU bound.  TFFT_NATIVE_V2 is set by hipcc only; the CPU emulation of the tests (g++) takes
// the scalar forms.
#if defined(__HIPCC__) && !defined(TFFT_SCALAR_COMPLEX)
#define TFFT_NATIVE_V2 1
typedef float tfft_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tfft_v2 nv(float2 a) { return tfft_v2{a.x, a.y}; }
__device__ __forceinline__ float2 f2(tfft_v2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return f2(nv(a) + nv(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return f2(nv(a) - nv(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    const tfft_v2 bs = tfft_v2{-b.y, b.x};                    // i*b
    return f2(a.x * nv(b) + a.y * bs);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return f2(nv(a) * s); }
__device__ __forceinline__ float2 cmuli(float2 a) { return f2(tfft_v2{-a.y, a.x}); }      // i*a
#else
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 cmuli(float2 a) { return make_float2(-a.y, a.x); }
#endif
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// Real-input FFT of length 2M through a complex FFT of length M on z[m] = x[2m] + i x[2m+1]:
//   forward: Z[k], Z[M-k], w = exp(+2 pi i k/2M)  ->  X[k] = Ev + w Od,  X[M-k] = conj(Ev - w Od)
//            Ev = (Z[k] + conj Z[M-k])/2,  Od = (Z[k] - conj Z[M-k])/(2i)
//   inverse: X[k], X[M-k], w                      ->  Z[k] = Ev + i Od,  Z[M-k] = conj(Ev - i Od)
//            Ev = (X[k] + conj X[M-k])/2,  Od = (X[k] - conj X[M-k])/2 * conj(w)
__device__ __forceinline__ void rsplit_fwd(float2 zk, float2 zm, float2 w, float2& xk, float2& xmk) {
    const float2 czm = cconj(zm);
    const float2 ev = cscale(cadd(zk, czm), 0.5f);
    const float2 od = cscale(cmuli(csub(zk, czm)), -0.5f);
    const float2 b = cmul(w, od);
    xk = cadd(ev, b);
    xmk = cconj(csub(ev, b));
}
__device__ __forceinline__ void rsplit_inv(float2 xk, float2 xm, float2 w, float2& zk, float2& zmk) {
    const float2 cxm = cconj(xm);
    const float2 ev = cscale(cadd(xk, cxm), 0.5f);
    const float2 iod = cmuli(cmul(cscale(csub(xk, cxm), 0.5f), cconj(w)));
    zk = cadd(ev, iod);
    zmk = cconj(csub(ev, iod));
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr int bitrev(int v, int bits) { return bits == 0 ? 0 : ((v & 1) << (bits - 1)) | bitrev(v >> 1, bits - 1); }
constexpr int imin(int a, int b) { return a < b ? a : b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }

// d * exp(SIGN * 2*pi*i * idx16/16), idx16 in 0..7 (compile-time after unrolling): alpha*d + beta*(SIGN*i*d)
template <int SIGN>
__device__ __forceinline__ float2 twmul16(float2 d, int idx16) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    constexpr float s = (float)SIGN;
#ifdef TFFT_NATIVE_V2
    const tfft_v2 dv = nv(d), rot = tfft_v2{-s * d.y, s * d.x};
    switch (idx16) {
        case 0: return d;
        case 1: return f2(C1 * dv + S1 * rot);
        case 2: return f2((dv + rot) * H);
        case 3: return f2(S1 * dv + C1 * rot);
        case 4: return f2(rot);
        case 5: return f2(C1 * rot - S1 * dv);
        case 6: return f2((rot - dv) * H);
        default: return f2(S1 * rot - C1 * dv);
    }
#else
    switch (idx16) {
        case 0: return d;
        case 1: return make_float2(d.x * C1 - s * d.y * S1, s * d.x * S1 + d.y * C1);
        case 2: return make_float2((d.x - s * d.y) * H, (s * d.x + d.y) * H);
        case 3: return make_float2(d.x * S1 - s * d.y * C1, s * d.x * C1 + d.y * S1);
        case 4: return make_float2(-s * d.y, s * d.x);
        case 5: return make_float2(-d.x * S1 - s * d.y * C1, s * d.x * C1 - d.y * S1);
        case 6: return make_float2((-d.x - s * d.y) * H, (s * d.x - d.y) * H);
        default: return make_float2(-d.x * C1 - s * d.y * S1, s * d.x * S1 - d.y * C1);
    }
#endif
}

// In-register radix-2 DIF network on v[OFF .. OFF+R): result X[q] lands in
// v[OFF + bitrev(q)].
template <int R, int SIGN, int OFF, int E>
struct DftReg {
    static __device__ __forceinline__ void run(float2 (&v)[E]) {
#pragma unroll
        for (int j = 0; j < R / 2; j++) {
            float2 a = v[OFF + j], b = v[OFF + j + R / 2];
            v[OFF + j] = cadd(a, b);
            v[OFF + j + R / 2] = twmul16<SIGN>(csub(a, b), j * (16 / R));
        }
        DftReg<R / 2, SIGN, OFF, E>::run(v);
        DftReg<R / 2, SIGN, OFF + R / 2, E>::run(v);
    }
};
template <int SIGN, int OFF, int E>
struct DftReg<1, SIGN, OFF, E> {
    static __device__ __forceinline__ void run(float2 (&)[E]) {}
};

// LDS layouts.  idx(n, b): element n of sequence b.
struct LayColumns {   // tile of NB adjacent columns, column index innermost
    int nb;
    __device__ __forceinline__ int idx(int n, int b) const { return n * nb + b; }
};
#ifndef TFFT_ROWS_SWZ
#define TFFT_ROWS_SWZ 1
#endif
#if TFFT_ROWS_SWZ
// Sequences back to back, element n at n ^ ((n >> 4) & 15): an XOR swizzle of the low four bits by the next four.
//   * the scatter of a radix-16 pass (16 contiguous lanes write elements 16 apart: ds_write_b64 groups of 16 lanes over 32 banks) lands on 16
//     distinct float2 columns, as the one-pad-slot-per-16 layout it replaces did;
//   * the gather (32 contiguous lanes read 32 consecutive elements: ds_read_b64 groups of 32 lanes over 64 banks) stays inside one
//     aligned run of 32 float2 = all 64 banks once.  With the pad slot, lane 31 of every group wrapped onto lane 0's banks and each gather
//     took two LDS cycles instead of one: SQ_LDS_BANK_CONFLICT was 34-41 % of the LDS cycles of the row and fused kernels (round 3).
struct LayRows {
    int pitch;
    __device__ __forceinline__ int idx(int n, int b) const { return b * pitch + (n ^ ((n >> 4) & 15)); }
    static constexpr int padded(int n) { return n < 16 ? 16 : n; }
};
#else
struct LayRows {      // sequences back to back, `pitch` float2 apart, one pad slot per 16 elements
    int pitch;
    __device__ __forceinline__ int idx(int n, int b) const { return b * pitch + n + (n >> 4); }
    static constexpr int padded(int n) { return n + (n >> 4) + 1; }
};
#endif

// exp(SIGN*2*pi*i*j/N) from a table tw[j] = exp(+2*pi*i*j/N)
template <int SIGN>
__device__ __forceinline__ float2 twload(const float2* __restrict__ tw, int j) {
    float2 w = tw[j];
    return SIGN > 0 ? w : cconj(w);
}

// Synchronisation between the scatter and the gather of an LDS exchange.
//   BlockSync: the sequence is spread over several waves -> workgroup barrier.
//   WaveSync : the whole sequence lives in ONE wave (N/E == 64).  DS instructions of a wave execute
//              in issue order, so only the compiler has to be kept from reordering / caching: no
//              s_barrier, waves of the workgroup run their passes independently.
// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a workgroup-scope release/acquire fence around
// s_barrier, and on gfx950 the release makes the compiler emit s_waitcnt vmcnt(0): EVERY global load in flight -- the prefetch of the
// next column tile, the twiddles fetched ahead -- was drained at the first exchange of a transform (tools/dma_probe.hip: a one-tile-ahead
// prefetch overlapped nothing, 0.47 ms where the raw barrier takes 0.31).  None of the transform kernels hands GLOBAL data from one
// thread of a workgroup to another, so the barrier only has to order LDS: wait for this wave's own LDS operations, s_barrier, and a
// compiler barrier on both sides.  (-DTFFT_RAW_BARRIER=0: the fenced form, for A/B runs; the CPU emulation always takes it.)
#ifndef TFFT_RAW_BARRIER
#define TFFT_RAW_BARRIER 1
#endif
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIPCC__) && TFFT_RAW_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    __syncthreads();
#endif
}
struct BlockSync { static __device__ __forceinline__ void sync() { lds_barrier(); } };
struct WaveSync {
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// Inter-pass twiddles.  Their table indices depend only on the thread's position (never on data), so a
// kernel fetches ALL of them into registers right after it has issued its data loads: the L2 latency
// of the table reads then overlaps the HBM latency of the data instead of stalling every pass.
template <int N, int E, int P>
struct TwCount {
    static constexpr int R = imin(E, N / P);
    static constexpr int mine = (P > 1) ? (E / R) * (R - 1) : 0;
    static constexpr int value = mine + TwCount<N, E, P * R>::value;
};
template <int N, int E>
struct TwCount<N, E, N> { static constexpr int value = 0; };
template <int N, int E>
constexpr int tw_regs() { return imax(1, TwCount<N, E, 1>::value); }

template <int N, int E, int SIGN, int P, int OFF, int TOT>
struct TwLoad {
    static constexpr int R = imin(E, N / P);
    static constexpr int T = N / E, Q = E / R;
    static __device__ __forceinline__ void run(float2 (&W)[TOT], int t, const float2* __restrict__ tw, int tws) {
        if constexpr (P > 1) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int k = (t + q * T) & (P - 1);
                const int base = k * (N / (P * R)) * tws;       // exp(SIGN*2*pi*i*r*k/(P*R)) = tw[r*base], r*k < P*R
#pragma unroll
                for (int r = 1; r < R; r++) W[OFF + q * (R - 1) + r - 1] = twload<SIGN>(tw, r * base);
            }
        }
        if constexpr (P * R < N) TwLoad<N, E, SIGN, P * R, OFF + ((P > 1) ? Q * (R - 1) : 0), TOT>::run(W, t, tw, tws);
    }
};
template <int N, int E, int SIGN, int TOT>
__device__ __forceinline__ void fft_prefetch_twiddles(float2 (&W)[TOT], int t, const float2* __restrict__ tw, int tws) {
    // tw[j*tws] = exp(+2*pi*i*j/N): one table per image dimension serves every sub-length
    if constexpr (N > 1) TwLoad<N, E, SIGN, 1, 0, TOT>::run(W, t, tw, tws);
}

// All radix passes of a length-N FFT for one thread.
//   u[m] holds x[t + m*T] on entry and X[t + m*T] on exit (T = N/E).
//   lds/lay: exchange buffer (N elements per sequence);  W: twiddles from fft_prefetch_twiddles.
//   Every thread of the workgroup must call this (it contains barriers), also
//   threads whose sequence is out of range.
template <int N, int E, int SIGN, int P, int OFF, int TOT, class Lay, class Sync>
struct FftPasses {
    static constexpr int R = imin(E, N / P);        // radix of this pass
    static constexpr bool LAST = (P * R == N);
    static constexpr int T = N / E;
    static constexpr int Q = E / R;                  // butterflies per thread
    static __device__ __forceinline__ void run(float2 (&u)[E], float2* lds, const Lay& lay, int t, int b,
                                               const float2 (&W)[TOT]) {
        float2 o[E];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            float2 v[R];
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = u[q + r * Q];
            const int i = t + q * T;
            const int k = i & (P - 1);
            if constexpr (P > 1) {
#pragma unroll
                for (int r = 1; r < R; r++) v[r] = cmul(v[r], W[OFF + q * (R - 1) + r - 1]);
            }
            DftReg<R, SIGN, 0, R>::run(v);
            if constexpr (LAST) {
#pragma unroll
                for (int s = 0; s < R; s++) o[q + s * Q] = v[bitrev(s, ilog2(R))];
            } else {
                const int j = (i - k) * R + k;
#pragma unroll
                for (int s = 0; s < R; s++) lds[lay.idx(j + s * P, b)] = v[bitrev(s, ilog2(R))];
            }
        }
        if constexpr (LAST) {
#pragma unroll
            for (int m = 0; m < E; m++) u[m] = o[m];
        } else {
            Sync::sync();
#pragma unroll
            for (int m = 0; m < E; m++) u[m] = lds[lay.idx(t + m * T, b)];
            Sync::sync();
            FftPasses<N, E, SIGN, P * R, OFF + ((P > 1) ? Q * (R - 1) : 0), TOT, Lay, Sync>::run(u, lds, lay, t, b, W);
        }
    }
};

template <int N, int E, int SIGN, class Sync = BlockSync, class Lay, int TOT>
__device__ __forceinline__ void fft_block(float2 (&u)[E], float2* lds, const Lay& lay, int t, int b,
                                          const float2 (&W)[TOT]) {
    if constexpr (N > 1) FftPasses<N, E, SIGN, 1, 0, TOT, Lay, Sync>::run(u, lds, lay, t, b, W);
}

// Same passes with the twiddles read from the table at the point of use (no register prefetch): for
// kernels whose occupancy is limited by VGPRs rather than by table latency.
//   active (uniform over the threads that share a sequence): false = this thread's sequence does not exist (a padded
//   row); it computes nothing and touches no LDS but still takes part in the workgroup barriers of BlockSync.
// Cross-lane form of ONE exchange (-DTFFT_XLANE=0: through LDS like the others): in the 1024-point transform of one wave
// (T = 64, E = 16) the exchange between the second radix-16 pass and the final radix-4 pass is a 4 x 4 transpose between the
// wave's four 16-lane rows and four registers -- lane (g, l) holds outputs s = a + 4q in registers, lane (r, l) needs register a = r of
// lane group g = ... -- which gfx950 does in registers with v_permlane16_swap / v_permlane32_swap (8 instructions per complex 4 x 4
// block) instead of 16 ds_write_b64 + 16 ds_read_b64 and two wave syncs.  Measured on one box (round 3, gpurun_out/r3n, 32 x 1080p):
// SQ_INSTS_LDS -18 .. -22 %, SQ_INSTS_VALU -2 .. -3 % (the address arithmetic of the LDS operations was more than the 32 swaps),
// fused forward 282.7 / 111.6 -> 282.1 / 107.3 us, fused inverse 472.9 -> 468.3 us, step 2.928 -> 2.919 and 3.030 -> 3.002 ms.
// (The FIRST exchange of that transform is a 16 x 16 transpose between the low lane nibble and the register index: ~128 swaps
// and selects, not tried; rows of two waves exchange across waves and need LDS anyway.)
#ifndef TFFT_XLANE
#define TFFT_XLANE 1
#endif
#if defined(__HIPCC__) && TFFT_XLANE
__device__ __forceinline__ void xl_swap16(float& x, float& y) {     // odd 16-lane rows of x <-> even rows of y
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xl_swap32(float& x, float& y) {     // upper 32 lanes of x <-> lower 32 lanes of y
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xl_transpose4(float2& a0, float2& a1, float2& a2, float2& a3) {
    xl_swap16(a0.x, a1.x); xl_swap16(a0.y, a1.y); xl_swap16(a2.x, a3.x); xl_swap16(a2.y, a3.y);
    xl_swap32(a0.x, a2.x); xl_swap32(a0.y, a2.y); xl_swap32(a1.x, a3.x); xl_swap32(a1.y, a3.y);
}
#define TFFT_XLANE_ON 1
#else
#define TFFT_XLANE_ON 0
#endif

template <int N, int E, int SIGN, int P, class Lay, class Sync>
struct FftPassesLazy {
    static constexpr int R = imin(E, N / P);
    static constexpr bool LAST = (P * R == N);
    static constexpr int T = N / E;
    static constexpr int Q = E / R;
    // the exchange after this pass done in registers across lanes (see above)
    static constexpr bool XL = TFFT_XLANE_ON && std::is_same<Sync, WaveSync>::value && N == 1024 && E == 16 && P == 16;
    static __device__ __forceinline__ void run(float2 (&u)[E], float2* lds, const Lay& lay, int t, int b,
                                               const float2* __restrict__ tw, int tws, bool active) {
        float2 o[E];
        if (active) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                float2 v[R];
#pragma unroll
                for (int r = 0; r < R; r++) v[r] = u[q + r * Q];
                const int i = t + q * T;
                const int k = i & (P - 1);
                if constexpr (P > 1) {
                    const int base = k * (N / (P * R)) * tws;
#pragma unroll
                    for (int r = 1; r < R; r++) v[r] = cmul(v[r], twload<SIGN>(tw, r * base));
                }
                DftReg<R, SIGN, 0, R>::run(v);
                if constexpr (LAST || XL) {
#pragma unroll
                    for (int s = 0; s < R; s++) o[q + s * Q] = v[bitrev(s, ilog2(R))];
                } else {
                    const int j = (i - k) * R + k;
#pragma unroll
                    for (int s = 0; s < R; s++) lds[lay.idx(j + s * P, b)] = v[bitrev(s, ilog2(R))];
                }
            }
        }
        if constexpr (LAST) {
            if (active) {
#pragma unroll
                for (int m = 0; m < E; m++) u[m] = o[m];
            }
        } else if constexpr (XL) {
#if TFFT_XLANE_ON
            if (active) {           // wave uniform: the wave is the row
#pragma unroll
                for (int q2 = 0; q2 < 4; q2++) xl_transpose4(o[4 * q2], o[4 * q2 + 1], o[4 * q2 + 2], o[4 * q2 + 3]);
#pragma unroll
                for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                    for (int r = 0; r < 4; r++) u[q2 + 4 * r] = o[r + 4 * q2];
            }
#endif
            FftPassesLazy<N, E, SIGN, P * R, Lay, Sync>::run(u, lds, lay, t, b, tw, tws, active);
        } else {
            Sync::sync();
            if (active) {
#pragma unroll
                for (int m = 0; m < E; m++) u[m] = lds[lay.idx(t + m * T, b)];
            }
            Sync::sync();
            FftPassesLazy<N, E, SIGN, P * R, Lay, Sync>::run(u, lds, lay, t, b, tw, tws, active);
        }
    }
};
template <int N, int E, int SIGN, class Sync = BlockSync, class Lay>
__device__ __forceinline__ void fft_block_lazy(float2 (&u)[E], float2* lds, const Lay& lay, int t, int b,
                                               const float2* __restrict__ tw, int tws, bool active = true) {
    if constexpr (N > 1) FftPassesLazy<N, E, SIGN, 1, Lay, Sync>::run(u, lds, lay, t, b, tw, tws, active);
}

// elements per thread for a length-N transform
constexpr int elems_for(int n) { return imin(16, n); }

}  // namespace tfft
