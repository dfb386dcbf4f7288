// mgm_post.hip -- the steps main() applies to the disparity maps right after the path:
//
//   median_filter      img_tools.h:203-238   NaN-aware, window clipped at the border, upper median v[n/2]
//   leftright_test     mgm.cc:68-91          |x - (Lx + R[Lx])| > tau, or Lx outside the other image  =>  NaN
//   back-projection    mgm.cc:433-443        v sampled at x + d (the reference's float index arithmetic), else u
//
// One thread per pixel; W*H work, nothing here is hot.  Default floating point (NaN-honouring).
#include "mgm_device.h"

namespace mgm {

// The reference gathers the non-NaN samples of the window and takes nth_element(n/2).  Selection without a
// sort: the k-th smallest is the sample with (#smaller) <= k < (#smaller + #equal).
__global__ void __launch_bounds__(256) k_median(const float *__restrict__ u, int nx, int ny, int nch, int radius,
                                                float *__restrict__ out)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * nch) return;
    const long long p = idx % npix;
    const int x = (int)(p % nx), y = (int)(p / nx);
    const float *pl = u + (idx / npix) * npix;
    const int x0 = x - radius < 0 ? 0 : x - radius, x1 = x + radius >= nx ? nx - 1 : x + radius;
    const int y0 = y - radius < 0 ? 0 : y - radius, y1 = y + radius >= ny ? ny - 1 : y + radius;
    int n = 0;
    for (int j = y0; j <= y1; j++)
        for (int i = x0; i <= x1; i++) {
            const float s = pl[i + (long long)j * nx];
            n += (s == s);
        }
    float res = pl[p];  // an all-NaN window leaves the pixel as it is
    if (n > 0) {
        const int k = n / 2;
        for (int j = y0; j <= y1; j++)
            for (int i = x0; i <= x1; i++) {
                const float s = pl[i + (long long)j * nx];
                if (!(s == s)) continue;
                int less = 0, equal = 0;
                for (int jj = y0; jj <= y1; jj++)
                    for (int ii = x0; ii <= x1; ii++) {
                        const float t = pl[ii + (long long)jj * nx];
                        less += (t < s);
                        equal += (t == s);
                    }
                if (less <= k && k < less + equal) res = s;
            }
    }
    out[idx] = res;
}

// Large windows (radius > 7: the kernel above is O(w^4) per pixel): the same order statistic by RADIX SELECTION -- floats
// mapped to keys whose unsigned order is the floats' order, the k-th smallest key found bit by bit from the top with one
// counting sweep of the window per bit (32 w^2 instead of w^4).
__device__ __forceinline__ unsigned order_key(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void __launch_bounds__(256) k_median_big(const float *__restrict__ u, int nx, int ny, int nch, int radius,
                                                    float *__restrict__ out, long long first, long long last)
{
    const long long npix = (long long)nx * ny;
    const long long idx = first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= last) return;
    const long long p = idx % npix;
    const int x = (int)(p % nx), y = (int)(p / nx);
    const float *pl = u + (idx / npix) * npix;
    const int x0 = x - radius < 0 ? 0 : x - radius, x1 = x + radius >= nx ? nx - 1 : x + radius;
    const int y0 = y - radius < 0 ? 0 : y - radius, y1 = y + radius >= ny ? ny - 1 : y + radius;
    int n = 0;
    for (int j = y0; j <= y1; j++)
        for (int i = x0; i <= x1; i++) {
            const float s = pl[i + (long long)j * nx];
            n += (s == s);
        }
    float res = pl[p];  // an all-NaN window leaves the pixel as it is
    if (n > 0) {
        int k = n / 2;
        unsigned prefix = 0;  // the bits of the answer above `b`
        for (int b = 31; b >= 0; b--) {
            int zeros = 0;  // samples that agree with the prefix and have bit b clear
            for (int j = y0; j <= y1; j++)
                for (int i = x0; i <= x1; i++) {
                    const float s = pl[i + (long long)j * nx];
                    const unsigned key = order_key(s);
                    const bool match = b == 31 ? true : (key >> (b + 1)) == prefix;
                    zeros += (s == s) && match && !((key >> b) & 1u);
                }
            if (k < zeros) prefix = prefix << 1;
            else {
                k -= zeros;
                prefix = (prefix << 1) | 1u;
            }
        }
        const unsigned ub = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
        res = __builtin_bit_cast(float, ub);
    }
    out[idx] = res;
}

hipError_t launch_median(const float *u, int nx, int ny, int nch, int radius, float *out, hipStream_t s)
{
    const long long n = (long long)nx * ny * nch;
    if (radius > 7) {
        // 33 sweeps of a (2r+1)^2 window per pixel, one thread per pixel: cut into launches of bounded work (~1.5e11 window
        // reads each, a fraction of a second) so that no single kernel runs long enough to look hung; the total is
        // capped by the caller (mgm_median_dev, kMedianMaxReads)
        const double per_pixel = 33.0 * (2.0 * radius + 1.0) * (2.0 * radius + 1.0);
        long long chunk = (long long)(1.5e11 / per_pixel);
        chunk = chunk < 256 ? 256 : (chunk / 256) * 256;
        for (long long a = 0; a < n; a += chunk) {
            const long long b = a + chunk < n ? a + chunk : n;
            hipLaunchKernelGGL(k_median_big, dim3((unsigned)((b - a + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, radius, out, a, b);
        }
    } else hipLaunchKernelGGL(k_median, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, radius, out);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_leftright(const float *__restrict__ dx, int nc, int nr, const float *__restrict__ Rdx,
                                                   int Rnc, float threshold, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nc * nr) return;
    const int x = (int)(i % nc), y = (int)(i / nc);
    const float d = dx[i];
    // round(x + d) converted to int; a NaN (or huge) disparity lands outside every image
    const double rr = __builtin_round((double)(x + d));
    const int Lx = (rr >= -2147483648.0 && rr <= 2147483647.0) ? (int)rr : -2147483647 - 1;
    float res = __builtin_nanf("");
    if (Lx < Rnc && Lx >= 0) {
        const float Rx = Lx + Rdx[Lx + (long long)y * Rnc];
        if (!(__builtin_fabs((double)(Rx - x)) > (double)threshold)) res = d;
    }
    out[i] = res;
}

hipError_t launch_leftright(const float *dx, int nc, int nr, const float *Rdx, int Rnc, float threshold, float *out,
                            hipStream_t s)
{
    const long long n = (long long)nc * nr;
    hipLaunchKernelGGL(k_leftright, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dx, nc, nr, Rdx, Rnc, threshold, out);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_backproject(const float *__restrict__ u, int nx, int ny, int nch,
                                                     const float *__restrict__ v, int vnx, int vny,
                                                     const float *__restrict__ disp, float *__restrict__ out)
{
    const int unpix = nx * ny, vnpix = vnx * vny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)unpix * nch) return;
    const int c = (int)(idx / unpix), p = (int)(idx % unpix);
    const int x = p % nx, y = p / nx;
    const float qx = disp[x + nx * y];
    const float px = x + qx, py = (float)y;
    const bool inside = px >= 0 && py >= 0 && px < vnx && py < vny;
    float r = u[x + y * nx + c * unpix];
    if (inside) {
        unsigned long long k = (unsigned long long)(x + qx + (y + 0.0f) * vnx + c * vnpix);  // (float arithmetic, as there)
        const unsigned long long last = (unsigned long long)vnpix * nch - 1;
        r = v[k < last ? k : last];
    }
    out[idx] = r;
}

hipError_t launch_backproject(const float *u, int nx, int ny, int nch, const float *v, int vnx, int vny, const float *disp,
                              float *out, hipStream_t s)
{
    const long long n = (long long)nx * ny * nch;
    hipLaunchKernelGGL(k_backproject, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, v, vnx, vny, disp, out);
    return hipGetLastError();
}

}  // namespace mgm
