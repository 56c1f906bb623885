// tools/dma_probe.hip -- what the column kernels can expect from LDS-DMA (global_load_lds_dwordx4) on the tile shape they stream:
// L rows x 128 B out of row-major planes (pitch M*8 B), one 512-thread workgroup per CU walking TPB adjacent tiles.
//   V0  register prefetch as k_fft_cols does it (16 x 8-byte loads per thread one tile ahead), consumed through LDS
//   V1  LDS-DMA into a second LDS buffer one tile ahead (8 x 1 KiB pieces per wave), consumed by 16 ds_read_b64 per thread
//   V2  V1 + 16 four-byte stores per thread per tile right after the compute (the |F|^2 store), waited for with vmcnt(0)
//   V3  V2 with the stores of tile i deferred to the start of tile i+1 (issued before the next DMA)
//   V5  V0 with raw barriers (s_waitcnt lgkmcnt(0) + s_barrier) instead of __syncthreads(), whose release fence drains vmcnt
//   V6  V1 + the 4-byte values of TWO adjacent tiles stored together as 128-byte row segments (8 B per lane after a swap with the
//       neighbouring lane), instead of two 64-byte segments per row
//   V7  V2 with the tiles of a row group dealt to PAIRS of workgroups on one XCD (ids 8 apart): one takes the even tiles, the other
//       the odd ones, so the two 64-byte halves of every 128-byte line are written at about the same time and can merge in the L2
//   V4  V2 with a COUNTED wait (vmcnt(16): the stores are the youngest 16 operations) -- also the in-order test of vmcnt
//       across loads, LDS-DMA and stores: its checksums must equal V1's
// `work` dummy FMA rounds per element stand in for the transform between the barriers.
// Prints GB/s of tile bytes read (+ written) and whether the per-workgroup checksums of all variants agree.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

constexpr int L = 512, C = 16, T = L / 16, NT = C * T, NW = NT / 64;

__device__ __forceinline__ void dma_tile(const float2* src, size_t pitch, unsigned char* buf, int w, int lane) {
#pragma unroll
    for (int i = 0; i < L / 8 / NW; i++) {
        const int ch = i * NW + w;
        const float2* s = src + (size_t)(ch * 8 + (lane >> 3)) * pitch + (lane & 7) * 2;
        // inline asm: the compiler must not know an LDS-DMA is pending, or it waits vmcnt(0) before the next ds_read (may-alias)
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(buf + ch * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
    }
}

template <int V>
__global__ void __launch_bounds__(NT) k_probe(const float2* __restrict__ in, float* __restrict__ out, float* __restrict__ sums, int M, int PH, int tpb, int work) {
    const int c = threadIdx.x, t = threadIdx.y, tid = t * C + c, w = tid >> 6, lane = tid & 63;
    const int g = blockIdx.y, plane = blockIdx.z;
    const size_t poff = (size_t)plane * PH * M + (size_t)g * L * M;
    int tile0 = blockIdx.x * tpb, tstep = 1;
    if (V == 7) {       // gridDim.x is a multiple of 16: blockIdx.x and blockIdx.x + 8 share an XCD (workgroups go round robin by linear id)
        const int bx = blockIdx.x, pair = (bx >> 4) * 8 + (bx & 7), par = (bx >> 3) & 1;
        tile0 = pair * 2 * tpb + par; tstep = 2;
    }
    float2* buf0 = reinterpret_cast<float2*>(smem);
    float2* buf1 = buf0 + L * C;
    float2 u[16], un[16];
    float acc = 0.f;
    float d[16];
    bool have_d = false;
    int dtile = 0;
    auto store_d = [&](int tile) {
        float* o = out + poff + (size_t)tile * C + c;
#pragma unroll
        for (int m = 0; m < 16; m++) o[(size_t)(t + m * T) * M] = d[m];
    };
    float dprev[16];
    auto bar = [&]() { if (V == 5) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } else __syncthreads(); };
    if (V == 0 || V == 5) {
#pragma unroll
        for (int m = 0; m < 16; m++) u[m] = in[poff + (size_t)(t + m * T) * M + tile0 * C + c];
    } else {
        dma_tile(in + poff + tile0 * C, M, (unsigned char*)buf0, w, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < tpb; i++) {
        const int tile = tile0 + i * tstep;
        float2* cur = (i & 1) ? buf1 : buf0;
        float2* nxt = (i & 1) ? buf0 : buf1;
        if (V == 3 && have_d) store_d(dtile);
        if (i + 1 < tpb) {
            if (V == 0 || V == 5) {
#pragma unroll
                for (int m = 0; m < 16; m++) un[m] = in[poff + (size_t)(t + m * T) * M + (tile + 1) * C + c];
            } else dma_tile(in + poff + (tile + tstep) * C, M, (unsigned char*)nxt, w, lane);
        }
        if (V == 0 || V == 5) {
#pragma unroll
            for (int m = 0; m < 16; m++) buf0[(t + m * T) * C + c] = u[m];
            bar();
#pragma unroll
            for (int m = 0; m < 16; m++) u[m] = buf0[((t + m * T) ^ 1) * C + c];
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) u[m] = cur[((t + m * T) ^ 1) * C + c];
        }
        for (int r = 0; r < work; r++) {
#pragma unroll
            for (int m = 0; m < 16; m++) { u[m].x = fmaf(u[m].x, 1.0001f, u[(m + 1) & 15].y); u[m].y = fmaf(u[m].y, 0.9999f, u[(m + 3) & 15].x); }
        }
#pragma unroll
        for (int m = 0; m < 16; m++) { acc += u[m].x + u[m].y; d[m] = u[m].x * u[m].x + u[m].y * u[m].y; }
        if (V == 2 || V == 4 || V == 7) store_d(tile);
        if (V == 3) { have_d = true; dtile = tile; }
        if (V == 6) {
            if (i & 1) {        // odd tile: this lane's value of the even tile (dprev) and of the odd tile (d); even lanes store the even tile's
                                // columns (c, c+1), odd lanes the odd tile's (c-1, c): one swap with the neighbouring lane per row
                float* o = out + poff + (size_t)(tile - 1) * C + (c & 1 ? C + c - 1 : c);
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const float send = (c & 1) ? dprev[m] : d[m];
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    const float2 v = (c & 1) ? make_float2(recv, d[m]) : make_float2(dprev[m], recv);
                    *reinterpret_cast<float2*>(o + (size_t)(t + m * T) * M) = v;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; m++) dprev[m] = d[m];
            }
        }
        if (V == 0 || V == 5) {
            bar();
#pragma unroll
            for (int m = 0; m < 16; m++) u[m] = un[m];
        } else {
            if (V == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (V == 3 && have_d) store_d(dtile);
    // checksum per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int k = 0; k < NT; k++) s += red[k]; sums[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s; }
}

template <int V>
static float run(const float2* in, float* out, float* sums, int M, int PH, int planes, int tpb, int work, int reps) {
    dim3 grid(M / C / tpb, PH / L, planes), block(C, T);
    const size_t lds = 2 * (size_t)L * C * sizeof(float2);
    CK(hipFuncSetAttribute((const void*)k_probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<V>, grid, block, lds, 0, in, out, sums, M, PH, tpb, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_probe<V>, grid, block, lds, 0, in, out, sums, M, PH, tpb, work);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int M = 2048, PH = 4096, planes = 24, tpb = argc > 1 ? atoi(argv[1]) : 16;
    const size_t n = (size_t)planes * PH * M;
    float2* in; float* out; float* sums;
    CK(hipMalloc(&in, n * sizeof(float2))); CK(hipMalloc(&out, n * sizeof(float)));
    const size_t nwg = (size_t)(M / C / tpb) * (PH / L) * planes;
    CK(hipMalloc(&sums, nwg * sizeof(float)));
    {
        std::vector<float2> h(n);
        unsigned s = 12345u;
        for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i].x = (float)((s >> 8) & 0xFFFF) / 65536.f - 0.5f; h[i].y = (float)(s >> 24) / 256.f - 0.5f; }
        CK(hipMemcpy(in, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    }
    const double rb = (double)n * 8, wb = (double)n * 4;
    std::vector<float> ref(nwg), got(nwg);
    for (int work : {0, 8, 16, 24, 32}) {
        float ms[8];
        if ((M / C / tpb) % 16 == 0) {
            ms[7] = run<7>(in, out, sums, M, PH, planes, tpb, work, 5);
            printf("work %2d | V7 paired workgroups, 64-B stores %.3f ms %.0f GB/s\n", work, ms[7], (rb + wb) / ms[7] / 1e6);
        }
        ms[5] = run<5>(in, out, sums, M, PH, planes, tpb, work, 5);
        ms[6] = run<6>(in, out, sums, M, PH, planes, tpb, work, 5);
        {   // V6 must write what V2 writes
            std::vector<float> o6((size_t)PH * M), o2((size_t)PH * M);
            CK(hipMemcpy(o6.data(), out, o6.size() * 4, hipMemcpyDeviceToHost));
            run<2>(in, out, sums, M, PH, planes, tpb, work, 1);
            CK(hipMemcpy(o2.data(), out, o2.size() * 4, hipMemcpyDeviceToHost));
            size_t badw = 0;
            for (size_t i = 0; i < o6.size(); i++) if (o6[i] != o2[i]) badw++;
            printf("work %2d | V5 regs+raw barrier %.3f ms %.0f GB/s | V6 paired 128-B stores %.3f ms %.0f GB/s  (plane 0: %zu values differ from V2's)\n", work, ms[5], rb / ms[5] / 1e6, ms[6], (rb + wb) / ms[6] / 1e6, badw);
        }
        ms[0] = run<0>(in, out, sums, M, PH, planes, tpb, work, 5);
        CK(hipMemcpy(got.data(), sums, nwg * sizeof(float), hipMemcpyDeviceToHost));
        ms[1] = run<1>(in, out, sums, M, PH, planes, tpb, work, 5);
        CK(hipMemcpy(ref.data(), sums, nwg * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad0 = 0;
        for (size_t i = 0; i < nwg; i++) if (ref[i] != got[i]) bad0++;
        if (bad0) printf("V0 vs V1: %zu mismatching checksums\n", bad0);
        ms[2] = run<2>(in, out, sums, M, PH, planes, tpb, work, 5);
        ms[3] = run<3>(in, out, sums, M, PH, planes, tpb, work, 5);
        ms[4] = run<4>(in, out, sums, M, PH, planes, tpb, work, 5);
        CK(hipMemcpy(got.data(), sums, nwg * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < nwg; i++) if (ref[i] != got[i]) bad++;
        printf("tpb %d work %2d | V0 regs %.3f ms %.0f GB/s | V1 dma %.3f ms %.0f GB/s | V2 dma+st %.3f ms %.0f GB/s | V3 deferred st %.3f ms %.0f GB/s | V4 counted %.3f ms %.0f GB/s  mismatching checksums %zu of %zu\n",
               tpb, work, ms[0], rb / ms[0] / 1e6, ms[1], rb / ms[1] / 1e6, ms[2], (rb + wb) / ms[2] / 1e6, ms[3], (rb + wb) / ms[3] / 1e6, ms[4], (rb + wb) / ms[4] / 1e6, bad, nwg);
    }
    return 0;
}
