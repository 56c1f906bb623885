// tools/membench.hip -- access-pattern microbenchmark used to choose the column-pass tiling
// (DESIGN.md "column pass").  Measures achieved GB/s (read + write bytes) of
//   copy      : contiguous float4 stream
//   tile S/R  : every workgroup copies a tile of R rows x S contiguous bytes out of a row-major
//               plane (pitch bytes per row), i.e. the access shape of a column FFT pass
// for a footprint beyond the 256 MiB Infinity Cache and for one that fits in it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// copy variants: U independent 16-byte loads in flight per thread, optional non-temporal loads / stores
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS>
__global__ void k_copy_v(const float4* __restrict__ in4, float4* __restrict__ out4, size_t n) {
    const v4f* in = reinterpret_cast<const v4f*>(in4);
    v4f* out = reinterpret_cast<v4f*>(out4);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * stride;
            if (j < n) v[u] = NTL ? __builtin_nontemporal_load(in + j) : in[j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * stride;
            if (j < n) { if (NTS) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u]; }
        }
    }
}

// write-only stream (the shape of the fused forward kernel's output: 8 bytes written per 1 read)
__global__ void k_write(float4* __restrict__ out, size_t n, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = make_float4(v, v, v, v);
}

// read-only stream (the shape of the median / capacity passes): U independent 16-byte loads per thread per
// iteration, grid-stride; the sum keeps the loads alive
template <int U>
__global__ void k_read(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (i + u * stride < n) ? in[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// LB = bytes per lane (8 or 16).  tile = blockIdx.x (column tile), blockIdx.y = row group, blockIdx.z = plane
template <int LB>
__global__ void k_tile(const char* __restrict__ in, char* __restrict__ out, int rows, size_t pitch, int seg,
                       int row_stride, int rows_per_block, size_t plane_bytes) {
    const int lanes_per_row = seg / LB;
    const int r0 = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    const int rstep = blockDim.x / lanes_per_row;
    const size_t base = (size_t)blockIdx.z * plane_bytes + (size_t)blockIdx.x * seg + (size_t)c * LB;
    const int g = blockIdx.y;   // row group: rows g + row_stride * j
    typedef typename std::conditional<LB == 16, float4, float2>::type V;
    for (int j0 = 0; j0 < rows_per_block; j0 += rstep * 4) {
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + r0 + u * rstep;
            const size_t row = (size_t)g + (size_t)row_stride * j;
            if (j < rows_per_block) v[u] = *reinterpret_cast<const V*>(in + base + row * pitch);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + r0 + u * rstep;
            const size_t row = (size_t)g + (size_t)row_stride * j;
            if (j < rows_per_block) *reinterpret_cast<V*>(out + base + row * pitch) = v[u];
        }
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  L2 %d MiB\n", p.gcnArchName, p.multiProcessorCount, p.l2CacheSize >> 20);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t big = (size_t)3 << 29;      // 1.5 GiB per buffer
    char *a, *b; CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big));
    CK(hipMemset(a, 1, big)); CK(hipMemset(b, 0, big));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) k_copy<<<2048, 256>>>((const float4*)a, (float4*)b, big / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep) printf("copy 1.5 GiB          : %7.1f GB/s\n", 5 * 2.0 * big / time_ms(e0, e1) / 1e6);
    }
    {
        const int grids[] = {1024, 2048, 4096, 8192};
        for (int g : grids) {
            float r[6];
            for (int v = 0; v < 6; v++) {
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 5; i++) {
                        if (v == 0) k_copy_v<1, false, false><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                        if (v == 1) k_copy_v<4, false, false><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                        if (v == 2) k_copy_v<4, false, true><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                        if (v == 3) k_copy_v<4, true, true><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                        if (v == 4) k_copy_v<8, false, true><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                        if (v == 5) k_copy_v<1, false, true><<<g, 256>>>((const float4*)a, (float4*)b, big / 16);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    r[v] = 5 * 2.0 * big / time_ms(e0, e1) / 1e6;
                }
            }
            printf("copy 1.5 GiB grid %5d x256: u1 %7.1f  u4 %7.1f  u4+nt-store %7.1f  u4+nt-both %7.1f  u8+nt-store %7.1f  u1+nt-store %7.1f GB/s\n", g, r[0], r[1], r[2], r[3], r[4], r[5]);
        }
    }
    for (int g : {1024, 2048, 4096, 8192}) {
        float r = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; i++) k_write<<<g, 256>>>((float4*)b, big / 16, (float)i);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            r = 5.0 * big / time_ms(e0, e1) / 1e6;
        }
        printf("write-only 1.5 GiB grid %5d x256: %7.1f GB/s\n", g, r);
    }
    {
        float* sink; CK(hipMalloc(&sink, 4));
        const int grids[] = {1024, 2048, 4096, 8192, 16384};
        for (int g : grids) {
            float r[3];
            for (int v = 0; v < 3; v++) {
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 5; i++) {
                        if (v == 0) k_read<1><<<g, 256>>>((const float4*)a, sink, big / 16);
                        if (v == 1) k_read<4><<<g, 256>>>((const float4*)a, sink, big / 16);
                        if (v == 2) k_read<8><<<g, 256>>>((const float4*)a, sink, big / 16);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    r[v] = 5.0 * big / time_ms(e0, e1) / 1e6;
                }
            }
            printf("read-only 1.5 GiB grid %5d x256: unroll 1 %7.1f  unroll 4 %7.1f  unroll 8 %7.1f GB/s\n", g, r[0], r[1], r[2]);
        }
    }
    // planes of 4096 rows x 16 KiB (= 2048 float2): 64 MiB per plane
    const int rows = 4096; const size_t pitch = 16384; const size_t plane = rows * pitch;
    struct Case { const char* name; int planes; int reps; } cases[] = {{"HBM  (24 planes, 1.5 GiB)", 24, 3}, {"MALL (1 plane, 64 MiB)   ", 1, 40}};
    for (auto& cs : cases) {
        printf("---- footprint %s\n", cs.name);
        for (int lb : {8, 16}) {
            for (int seg : {32, 64, 128, 256, 512}) {
                if (seg < lb * 2) continue;
                // (a) tall tile: all 4096 rows, consecutive;  (b) two-step shape: 64 rows at stride 64, 64 groups
                for (int shape = 0; shape < 2; shape++) {
                    const int rpb = shape == 0 ? rows : 64, rstride = shape == 0 ? 1 : 64, groups = shape == 0 ? 1 : 64;
                    dim3 grid((unsigned)(pitch / seg), groups, cs.planes);
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        CK(hipEventRecord(e0));
                        for (int i = 0; i < cs.reps; i++) {
                            if (lb == 8) k_tile<8><<<grid, 256>>>(a, b, rows, pitch, seg, rstride, rpb, plane);
                            else k_tile<16><<<grid, 256>>>(a, b, rows, pitch, seg, rstride, rpb, plane);
                        }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms = time_ms(e0, e1) / cs.reps; if (ms < best) best = ms;
                    }
                    printf("lane %2d B  seg %3d B  %s : %7.1f GB/s  (%.1f us)\n", lb, seg,
                           shape == 0 ? "tall 4096 rows     " : "64 rows @ stride 64", 2.0 * plane * cs.planes / best / 1e6, best * 1e3);
                }
            }
        }
    }
    return 0;
}
