// Development aid: what a HIP process pays before its first kernel on this box (hipInit, stream, allocations), next to the CLI one-shot timings
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <unistd.h>
__global__ void k(int *p) { p[0] = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double t0 = now();
    int n = 0;
    hipGetDeviceCount(&n);
    double t1 = now();
    hipSetDevice(0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    double t2 = now();
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t3 = now();
    void *h;
    hipHostMalloc(&h, 64, hipHostMallocDefault);
    double t4 = now();
    int *d;
    hipMalloc((void **)&d, 1 << 20);
    double t5 = now();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    double t6 = now();
    void *big;
    hipMalloc(&big, (size_t)34 << 30);
    double t7 = now();
    printf("count %.1f  setdevice+props %.1f  stream %.1f  hostmalloc %.1f  malloc1M %.1f  first kernel %.1f  malloc34G %.1f | total %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3,
           t5 - t4, t6 - t5, t7 - t6, t7 - t0);
    fflush(stdout);
    _exit(0);
}
