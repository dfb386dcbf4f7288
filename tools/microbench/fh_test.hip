// Development aid: fh_minconv of mgm_pass_common.h against the sequential recurrence on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-honor-nans -I mgm_amd/csrc -I include tools/microbench/fh_test.hip -o /tmp/fh_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mgm_device.h"
#include "mgm_pass_common.h"
using namespace mgm;

template <int LPL>
__global__ void k(const float *in, float *out, unsigned *sw, float P1, float P2, int n)
{
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    float M[LPL];
    for (int q = 0; q < LPL; q++) M[q] = in[(size_t)w * LPL * 64 + lane * LPL + q];
    float m = M[0];
    for (int q = 1; q < LPL; q++) m = fminf(m, M[q]);
    m = wave_min(m);
    unsigned s = 0;
    fh_minconv<LPL, true>(M, m, P1, P2, lane, LPL * 64, s);
    for (int q = 0; q < LPL; q++) out[(size_t)w * LPL * 64 + lane * LPL + q] = M[q];
    if (lane == 0) sw[w] = s;
}

int main()
{
    constexpr int LPL = 4, L = 256;
    const int n = 20000;
    std::vector<float> h((size_t)n * L), ref((size_t)n * L), got((size_t)n * L);
    srand(1);
    for (int i = 0; i < n; i++) {
        const int kind = i % 4;
        const int inf_lo = (kind == 1) ? rand() % 250 : 0, inf_hi = (kind == 2) ? L - rand() % 250 : L;
        for (int o = 0; o < L; o++) {
            float v = (float)(rand() % 97) + (float)(rand() % 3000) / 3.0f * (kind == 3 ? 0.01f : 1.0f);
            if (rand() % 50 == 0) v = (float)(rand() % 7) / 3.0f;
            h[(size_t)i * L + o] = (o < inf_lo || o >= inf_hi) ? INFINITY : v;
        }
    }
    float *din, *dout;
    unsigned *dsw;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, h.size() * 4);
    hipMalloc(&dsw, n * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (float P1 : {2.0f, 8.0f, 3.0f, 2.5f, 0.7f}) {
        const float P2 = 20000.0f;
        hipLaunchKernelGGL(k<LPL>, dim3((n + 3) / 4), dim3(256), 0, 0, din, dout, dsw, P1, P2, n);
        hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> sw(n);
        hipMemcpy(sw.data(), dsw, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, rep = 0, tot = 0;
        unsigned worst = 0;
        for (int i = 0; i < n; i++) {
            float *M = &ref[(size_t)i * L];
            float m = INFINITY;
            for (int o = 0; o < L; o++) { M[o] = h[(size_t)i * L + o]; m = fminf(m, M[o]); }
            for (int o = 1; o < L; o++) { volatile float t = M[o - 1] + P1; M[o] = fminf(M[o], t); }
            for (int o = L - 2; o >= 0; o--) { volatile float t = M[o + 1] + P1; M[o] = fminf(M[o], t); }
            for (int o = 0; o < L; o++) { M[o] = fminf(M[o], m + P2); bad += !(M[o] == got[(size_t)i * L + o]); }
            rep += sw[i] > 2; tot += sw[i]; worst = sw[i] > worst ? sw[i] : worst;
        }
        printf("P1 %.2f: mismatching labels %ld, slabs repaired %.2f%%, sweeps/slab %.3f, worst %u\n", P1, bad, 100.0 * rep / n, (double)tot / n, worst);
    }
    return 0;
}
