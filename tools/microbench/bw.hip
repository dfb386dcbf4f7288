// HBM bandwidth micro-benchmark (development aid): streaming read, write and copy with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_write(float4 *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += st) p[i] = v;
}
__global__ void k_read(const float4 *p, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float a = 0;
    for (; i < n; i += st) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) *out = a;
}
__global__ void k_copy(const float4 *s, float4 *d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) d[i] = s[i];
}
// one wave writes 1 KiB contiguous per iteration, 15 "lines" per workgroup far apart (K3-like)
__global__ void k_write_lines(float4 *p, size_t line_elems, int steps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *q = p + ((size_t)blockIdx.x * 16 + wave) * line_elems + lane;
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (int s = 0; s < steps; s++) q[(size_t)s * 64] = v;
}
// k_wta-like: every wave sums NS streams (volumes `vol` floats4 apart), CH KiB contiguous per stream per iteration
template <int NS, int CH>
__global__ void k_read_streams(const float4 *p, size_t vol, size_t npix, float *out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    float a = 0;
    for (size_t px = wave * CH; px < npix; px += nw * CH) {
        float4 v[NS][CH];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int c = 0; c < CH; c++) v[s][c] = p[s * vol + (px + c) * 64 + lane];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int c = 0; c < CH; c++) a += v[s][c].x + v[s][c].y + v[s][c].z + v[s][c].w;
    }
    if (a == 123.456f) *out = a;
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    float4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, double gb, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-28s %8.3f ms  %8.1f GB/s\n", name, best, gb / (best * 1e-3));
    };
    for (int blocks : {2048, 8192}) {
        printf("grid %d x 256\n", blocks);
        timeit("write  8 GiB", 8.59, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, n); });
        timeit("read   8 GiB", 8.59, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, o); });
        timeit("copy   8+8 GiB", 17.18, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    // 256 workgroups x 16 waves, each wave streams its own 2 MiB "line" 1 KiB at a time
    timeit("line writes 256x16x2MiB", 256.0 * 16 * 2048 * 1024 / 1e9,
           [&] { hipLaunchKernelGGL(k_write_lines, dim3(256), dim3(1024), 0, 0, a, (size_t)131072, 2048); });
    timeit("line writes 512x16x1MiB", 512.0 * 16 * 1024 * 1024 / 1e9,
           [&] { hipLaunchKernelGGL(k_write_lines, dim3(512), dim3(1024), 0, 0, a, (size_t)65536, 1024); });
    {   // 8 streams of 1 GiB inside the 8 GiB buffer `a`
        const size_t vol = ((size_t)1 << 30) / 16, npix = vol / 64;
        for (int blocks : {1024, 2048, 4096, 8192}) {
            printf("grid %d x 256\n", blocks);
            timeit("8 streams, 1 KiB/iter", 8.59, [&] { hipLaunchKernelGGL((k_read_streams<8, 1>), dim3(blocks), dim3(256), 0, 0, a, vol, npix, o); });
            timeit("8 streams, 2 KiB/iter", 8.59, [&] { hipLaunchKernelGGL((k_read_streams<8, 2>), dim3(blocks), dim3(256), 0, 0, a, vol, npix, o); });
            timeit("8 streams, 4 KiB/iter", 8.59, [&] { hipLaunchKernelGGL((k_read_streams<8, 4>), dim3(blocks), dim3(256), 0, 0, a, vol, npix, o); });
        }
    }
    {   // 8 streams 2.1 GB apart (the real Lr volume stride of a 1920x1080x256 run): 17 GB footprint
        float4 *big;
        const size_t volb = (size_t)1920 * 1080 * 256 * 4, vol = volb / 16, npix = (size_t)1920 * 1080;
        if (hipMalloc(&big, volb * 8) == hipSuccess) {
            hipMemset(big, 0, volb * 8);
            for (int blocks : {2048, 4096}) {
                printf("grid %d x 256, 8 x 2.1 GB volumes\n", blocks);
                timeit("8 streams, 1 KiB/iter", 8 * volb / 1e9, [&] { hipLaunchKernelGGL((k_read_streams<8, 1>), dim3(blocks), dim3(256), 0, 0, big, vol, npix, o); });
                timeit("8 streams, 2 KiB/iter", 8 * volb / 1e9, [&] { hipLaunchKernelGGL((k_read_streams<8, 2>), dim3(blocks), dim3(256), 0, 0, big, vol, npix, o); });
                timeit("8 streams, 4 KiB/iter", 8 * volb / 1e9, [&] { hipLaunchKernelGGL((k_read_streams<8, 4>), dim3(blocks), dim3(256), 0, 0, big, vol, npix, o); });
            }
        }
    }
    return 0;
}
