// Store-pattern micro-benchmark (development aid): the Lr write pattern of K3 without any of its compute.
// 512 workgroups ("bands", two per CU) x 15 waves ("lines"); every wave writes one slab per step, either along a
// row of its volume (row passes: consecutive slabs) or down a column (column passes: one image row apart).
// SUBV=2 splits the wave's 1 KiB store between two volumes as the 128-label kernels do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Geo { long long volb, rowb, slab; int W, H, nvol, subv, column; };
// the same stores in lock-step: one s_barrier per step (all 16 waves) and `delay` x 64 clocks of s_sleep standing
// in for the step's compute
__global__ __launch_bounds__(1024) void k_lockstep(char *base, Geo g, int steps, int delay, int phases)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lanes = 64 / g.subv, grp = lane / lanes, l = lane % lanes;
    const int slots = g.nvol / g.subv;
    const int vol = (blockIdx.x % slots) * g.subv + grp;
    const int band = blockIdx.x / slots;
    const long long line = (long long)band * 15 + wave;
    char *p = base + vol * g.volb + (g.column ? line * g.slab : line * g.rowb) + l * 16;
    const long long stride = g.column ? g.rowb : g.slab;
    float4 v = {1.f, 2.f, 3.f, 4.f};
    const int n = g.column ? (steps < g.H ? steps : g.H) : (steps < g.W ? steps : g.W);
    const bool live = wave < 15 && line < (g.column ? g.W : g.H);
    const bool mutate = phases < 0;  // overwrite the stored registers right after the store (WAR on the store data)
    phases = phases < 0 ? -phases : phases;
    for (int s = 0; s < n; s++) {
        // the store sits at a wave-dependent point of the step's "compute" when phases > 1
        const int before = phases > 1 ? delay * (wave % phases) / phases : delay;
        for (int d = 0; d < before; d++) __builtin_amdgcn_s_sleep(1);
        if (live) *(float4 *)(p + s * stride) = v;
        if (mutate) {
            v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        }
        for (int d = before; d < delay; d++) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_s_barrier();
    }
}
__global__ __launch_bounds__(1024) void k_pattern(char *base, Geo g, int steps)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 15) return;
    const int lanes = 64 / g.subv, grp = lane / lanes, l = lane % lanes;
    const int slots = g.nvol / g.subv;               // volume groups sharing the launch
    const int vol = (blockIdx.x % slots) * g.subv + grp;
    const int band = blockIdx.x / slots;
    const long long line = (long long)band * 15 + wave;
    char *p = base + vol * g.volb + (g.column ? line * g.slab : line * g.rowb) + l * 16;
    const long long stride = g.column ? g.rowb : g.slab;
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    const int n = g.column ? (steps < g.H ? steps : g.H) : (steps < g.W ? steps : g.W);
    if (line >= (g.column ? g.W : g.H)) return;
    for (int s = 0; s < n; s++) *(float4 *)(p + s * stride) = v;
}
int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 128, nvol = argc > 2 ? atoi(argv[2]) : 16, W = 1920, H = 1080;
    const int subv = 256 / L;
    Geo g{(long long)W * H * L * 4, (long long)W * L * 4, (long long)L * 4, W, H, nvol, subv, 0};
    char *a;
    if (hipMalloc(&a, g.volb * nvol) != hipSuccess) return 1;
    (void)hipMemset(a, 0, g.volb * nvol);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int column = 0; column < 2; column++)
        for (int blocks : {256, 512, 1024}) {
            g.column = column;
            const int slots = nvol / subv, lines = column ? W : H, bands = (lines + 14) / 15;
            const int nb = blocks < bands * slots ? blocks : bands * slots;
            const int steps = column ? H : W;
            float best = 1e9;
            for (int r = 0; r < 4; r++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(1024), 0, 0, a, g, steps);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)nb * 15 * steps * 64 * 16;
            printf("L=%d nvol=%d %s pattern, %4d bands: %7.3f ms %8.1f GB/s\n", L, nvol, column ? "column" : "row   ", nb, best, bytes / best / 1e6);
        }
    // lock-step variant, 512 bands (two per CU), row pattern
    g.column = 0;
    for (int phases : {1, -1, -3})
    for (int delay : {0, 16, 32, 48}) {
        for (int stores = 1; stores >= 0; stores--) {
            const int nb = 512, steps = stores ? W : -1;
            float best = 1e9;
            for (int r = 0; r < 3; r++) {
                (void)hipEventRecord(e0);
                if (stores) hipLaunchKernelGGL(k_lockstep, dim3(nb), dim3(1024), 0, 0, a, g, W, delay, phases);
                else { Geo g0 = g; g0.W = W; hipLaunchKernelGGL(k_lockstep, dim3(nb), dim3(1024), 0, 0, a + g.volb * nvol, g0, W, delay, phases); }
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            (void)steps;
            const double bytes = (double)nb * 15 * W * 64 * 16;
            printf("lock-step, %2d store phases, delay %2d: %7.3f ms (%.3f us/step) %8.1f GB/s\n", phases, delay, best, best * 1e3 / W, bytes / best / 1e6);
            break;  // (the store-free leg is the delay alone: see delay rows against the 0 row)
        }
    }
    return 0;
}
