#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mgm_device.h"
#include "mgm_pass_common.h"
using namespace mgm;
template <int LPL>
__global__ void k(const float *in, float *ca, float *cr, float P1, int n)
{
    const int lane = threadIdx.x & 63;
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    float M[LPL];
    for (int q = 0; q < LPL; q++) M[q] = in[(size_t)w * LPL * 64 + lane * LPL + q];
    float a = M[0];
    for (int q = 1; q < LPL; q++) a = fminf(M[q], a + P1);
    ca[w * 64 + lane] = a;
    cr[w * 64 + lane] = fh_repair<LPL, true>(a, P1, lane);
}
int main()
{
    constexpr int LPL = 4, L = 256;
    const int n = 2000;
    std::vector<float> h((size_t)n * L);
    srand(1);
    for (int i = 0; i < n; i++)
        for (int o = 0; o < L; o++) {
            float v = (float)(rand() % 97) + (float)(rand() % 3000) / 3.0f;
            if (rand() % 50 == 0) v = (float)(rand() % 7) / 3.0f;
            h[(size_t)i * L + o] = v;
        }
    float *din, *da, *dr;
    hipMalloc(&din, h.size() * 4); hipMalloc(&da, n * 64 * 4); hipMalloc(&dr, n * 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const float P1 = 2.0f;
    hipLaunchKernelGGL(k<LPL>, dim3((n + 3) / 4), dim3(256), 0, 0, din, da, dr, P1, n);
    std::vector<float> a(n * 64), r(n * 64);
    hipMemcpy(a.data(), da, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, r.size() * 4, hipMemcpyDeviceToHost);
    int shown = 0; long bad = 0;
    for (int i = 0; i < n; i++) {
        std::vector<float> M(h.begin() + (size_t)i * L, h.begin() + (size_t)(i + 1) * L);
        for (int o = 1; o < L; o++) { volatile float t = M[o - 1] + P1; M[o] = fminf(M[o], t); }
        for (int l = 0; l < 64; l++) {
            const float t = M[l * LPL + LPL - 1];
            if (t != r[i * 64 + l]) {
                bad++;
                if (shown++ < 12) printf("slab %d lane %d: a %.9g repaired %.9g true %.9g (a[l-1] %.9g)\n", i, l, a[i * 64 + l], r[i * 64 + l], t, l ? a[i * 64 + l - 1] : 0.f);
            }
        }
    }
    printf("bad carries %ld of %d\n", bad, n * 64);
    return 0;
}
