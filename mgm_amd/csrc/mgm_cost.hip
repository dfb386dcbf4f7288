// mgm_cost.hip -- K1 census transform, K2 cost-volume fill, K7 edge weights.
//
//   K1  census_transform          census_tools.cc:16-57, 76-99, 127-153
//   K2  allocate_and_fill_sgm_costvolume   mgm_costvolume.h:390-422
//       with computeC_AD (23-33), computeC_SD (34-44),
//       computeC_census_on_preprocessed_images (65-78)
//   K7  compute_mgm_weights       mgm_weights.h:26-85
//
// Compiled with default (NaN-honouring) floating point: non-finite pixels and
// costs follow IEEE rules exactly as on the CPU.
#include "mgm_cost_common.h"

namespace mgm {

// ---- K1 -----------------------------------------------------------------------
// One thread per pixel.  Bit order: channel, dy, dx, centre skipped; bit =
// (centre < neighbour), 0 when the neighbour is outside the image (NaN sample,
// census_tools.cc:28-33); bits packed MSB-first into bytes (16-25), bytes laid
// little-endian into 32-bit words (the reference memcpy's them into floats).
__global__ void __launch_bounds__(256) k_census(const float *__restrict__ u, int nx, int ny, int nch, int wr,
                                                int nwords, uint32_t *__restrict__ out)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const int x = (int)(idx % nx), y = (int)(idx / nx);
    uint32_t word = 0;   // word being assembled
    uint32_t byte = 0;   // byte being assembled
    int nbit = 0, nbyte = 0, w = 0;
    for (int l = 0; l < nch; l++) {
        const float *pl = u + (long long)l * npix;
        const float a = pl[idx];
        for (int j = -wr; j <= wr; j++)
            for (int i = -wr; i <= wr; i++) {
                if (!i && !j) continue;
                const int xx = x + i, yy = y + j;
                uint32_t bit = 0;
                if (xx >= 0 && xx < nx && yy >= 0 && yy < ny) bit = a < pl[(long long)yy * nx + xx];
                byte = byte * 2 + bit;
                if (++nbit == 8) {
                    word |= byte << (8 * nbyte);
                    byte = 0;
                    nbit = 0;
                    if (++nbyte == 4) {
                        out[idx + (long long)w * npix] = word;
                        word = 0;
                        nbyte = 0;
                        w++;
                    }
                }
            }
    }
    if (nbyte) out[idx + (long long)w * npix] = word;
    (void)nwords;
}

hipError_t launch_census(const float *u, int nx, int ny, int nch, int winradius, uint32_t *out, hipStream_t s)
{
    const long long npix = (long long)nx * ny;
    const int side = 2 * winradius + 1;
    const int nwords = (nch * (side * side - 1) / 8 + 3) / 4;
    hipLaunchKernelGGL(k_census, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, winradius,
                       nwords, out);
    return hipGetLastError();
}

// ---- prefilters of the non-census costs ------------------------------------------
// apply_filter (img_tools.h:105-127) with a small 2-D kernel: Neumann boundary (the nearest pixel), products
// accumulated in row-major kernel order in fp32, exactly as the reference's loops do.  One thread per output
// sample.  Serves sobelx (3x3) and, called twice, the separable gblur (1 x r, then r x 1; img_tools.h:140-180).
struct FilterTaps {
    float f[39];
};
__global__ void __launch_bounds__(256) k_filter2d(const float *__restrict__ u, int nx, int ny, int nch, const FilterTaps K,
                                                  int fnx, int fny, float *__restrict__ out)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * nch) return;
    const int c = (int)(idx / npix);
    const long long p = idx % npix;
    const int x0 = (int)(p % nx) - fnx / 2, y0 = (int)(p / nx) - fny / 2;  // top-left tap
    const float *pl = u + (long long)c * npix;
    float v = 0;
    for (int t = 0; t < fnx * fny; t++) {  // taps in row-major order, one rounded product + one rounded add each
        const int x = min(max(x0 + t % fnx, 0), nx - 1);  // Neumann boundary: the nearest pixel
        const int y = min(max(y0 + t / fnx, 0), ny - 1);
        v += pl[x + (long long)y * nx] * K.f[t];
    }
    out[idx] = v;
}

hipError_t launch_filter2d(const float *u, int nx, int ny, int nch, const float *taps, int fnx, int fny, float *out,
                           hipStream_t s)
{
    if (fnx * fny > 39) return hipErrorInvalidValue;
    FilterTaps K;
    for (int i = 0; i < fnx * fny; i++) K.f[i] = taps[i];
    const long long n = (long long)nx * ny * nch;
    hipLaunchKernelGGL(k_filter2d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, K, fnx, fny, out);
    return hipGetLastError();
}

// ---- the costs that look at more than one sample per image (general kernel) ------------
__device__ __forceinline__ float btad1(const float *__restrict__ pu, int nx, int px, const float *__restrict__ pv, int vnx, int qx)
{
    const BtSpan a = bt_span(pu, nx, px), b = bt_span(pv, vnx, qx);
    const float a_to_b = tri_high(0.0f, a.centre - b.hi, b.lo - a.centre);
    const float b_to_a = tri_high(0.0f, b.centre - a.hi, a.lo - b.centre);
    return __builtin_fabsf(a_to_b < b_to_a ? a_to_b : b_to_a);
}

// computeC_clippedNCC (mgm_costvolume.h:137-165): window sums in float, the normalisation in double (0.0000001 and
// sqrt are doubles), a window sample outside either image or NaN => INFINITY.
__device__ __forceinline__ float cost_ncc(const CostParams &P, int px, int py, int qx, int qy)
{
    const long long npix = (long long)P.nx * P.ny, vpix = (long long)P.vnx * P.vny;
    const int nch = P.nch, hw = P.hwin;
    float NCC = 0;
    for (int t = 0; t < nch; t++) {
        float mu1 = 0, mu2 = 0, s1 = 0, s2 = 0, prod = 0;
        int n = 0;
        for (int i = -hw; i <= hw; i++)
            for (int j = -hw; j <= hw; j++) {
                const int ax = px + i, ay = py + j, bx = qx + i, by = qy + j;
                if (ax < 0 || ay < 0 || ax >= P.nx || ay >= P.ny || bx < 0 || by < 0 || bx >= P.vnx || by >= P.vny)
                    return __builtin_huge_valf();
                const float v1 = P.u[ax + (long long)ay * P.nx + t * npix];
                const float v2 = P.v[bx + (long long)by * P.vnx + t * vpix];
                if (!(v1 == v1) || !(v2 == v2)) return __builtin_huge_valf();
                mu1 += v1;
                mu2 += v2;
                s1 += v1 * v1;
                s2 += v2 * v2;
                prod += v1 * v2;
                n++;
            }
        mu1 /= n;
        mu2 /= n;
        s1 /= n;
        s2 /= n;
        prod /= n;
        const float var = (s1 - mu1 * mu1) * (s2 - mu2 * mu2);
        const double den = (0.0000001 > var) ? 0.0000001 : (double)var;
        NCC = (float)(NCC + (prod - mu1 * mu2) / __builtin_sqrt(den));
    }
    const float m = (NCC < nch) ? NCC : (float)nch;
    const float c = (0 > m) ? 0 : m;
    const float clipped = nch - c;
    return clipped * 64;
}

// ---- K2 -----------------------------------------------------------------------
// One wavefront per pixel; lane l fills labels o = l, l+64, ... so that every
// store instruction writes 64 consecutive floats of the pixel's slab.
__global__ void __launch_bounds__(256) k_cost(const CostParams P)
{
    const long long npix = (long long)P.nx * P.ny;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const int lane = threadIdx.x & 63;
    const int x = (int)(pix % P.nx), y = (int)(pix / P.nx);
    const long long vpix = (long long)P.vnx * P.vny;
    float *Cp = P.C ? P.C + pix * P.L : nullptr;  // nullptr: only the compact copy is wanted
    uint8_t *Cp8 = P.C8 ? P.C8 + pix * P.L * P.cbytes : nullptr;
    const bool two = P.cbytes == 2;  // (the compact copy holds two bytes per cost)
    const bool yin = (y < P.vny);  // q.y = p.y >= 0 always
    bool anyfinite = false, bad8 = false, nanv = false;
    int rl = 0, rh = P.L - 1;  // the pixel's own label range (ragged volumes)
    if (P.rlo) {
        rl = (int)P.rlo[pix] - P.dmin;
        rh = (int)P.rhi[pix] - P.dmin;
    }
    for (int o = lane; o < P.L; o += 64) {
        const int qx = x + o + P.dmin;
        float e = P.trunc;
        if (o < rl || o > rh) {  // not a label of this pixel
            if (Cp) Cp[o] = __builtin_huge_valf();
            if (Cp8) {
                if (two) reinterpret_cast<unsigned short *>(Cp8)[o] = 65535;
                else Cp8[o] = 255;
            }
            continue;
        }
        if (yin && qx >= 0 && qx < P.vnx) {
            const long long q = (long long)y * P.vnx + qx;
            if (P.costfn == 2) {
                float r = 0;
                for (int t = 0; t < P.nch; t++) {
                    const uint32_t xr = P.cu[pix + (long long)t * npix] ^ P.cv[q + (long long)t * vpix];
                    r += (float)__builtin_popcount(xr);
                }
                e = (float)((double)r * 1.0 / (double)P.nch);
            } else if (P.costfn == 3) {
                e = cost_ncc(P, x, y, qx, y);
            } else if (P.costfn >= 4) {  // computeC_BTAD / computeC_BTSD (mgm_costvolume.h:114-135)
                float val = 0;
                for (int t = 0; t < P.nch; t++) {
                    const float b = btad1(P.u + (long long)t * npix + (long long)y * P.nx, P.nx, x,
                                          P.v + (long long)t * vpix + (long long)y * P.vnx, P.vnx, qx);
                    val += (P.costfn == 5) ? b * b : b;
                }
                e = val;
            } else {
                float tmp = 0;
                for (int t = 0; t < P.nch; t++) {
                    float d = P.u[pix + (long long)t * npix] - P.v[q + (long long)t * vpix];
                    d = (d > -d) ? d : -d;
                    if (P.costfn == 0) tmp += d;
                    else tmp += d * d;
                }
                e = tmp;
            }
        }
        e = (e < P.trunc) ? e : P.trunc;
        if (Cp) Cp[o] = e;
        anyfinite |= finite_bits(e);
        nanv |= e != e;  // (only a NaN truncDist gets here: a NaN cost loses the comparison above)
        if (Cp8) {
            if (two) {
                const unsigned b = c16_encode(e);
                bad8 |= b > 65535u;
                reinterpret_cast<unsigned short *>(Cp8)[o] = (unsigned short)b;
            } else {
                const unsigned b = c8_encode(e);
                bad8 |= b > 255u;
                Cp8[o] = (uint8_t)b;
            }
        }
    }
    // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
    if (__builtin_amdgcn_ballot_w64(anyfinite) == 0ull)
        for (int o = lane; o < P.L; o += 64) {
            if (o < rl || o > rh) continue;
            if (Cp) Cp[o] = 0.0f;
            if (Cp8) {
                if (two) reinterpret_cast<unsigned short *>(Cp8)[o] = 0;
                else Cp8[o] = 0;
            }
        }
    else if (Cp8 && __builtin_amdgcn_ballot_w64(bad8) != 0ull && lane == 0)
        flag_once(P.bad8, 1u);
    if (P.bad8 && __builtin_amdgcn_ballot_w64(nanv) != 0ull && lane == 0) flag_once(P.bad8, 2u);
}

// compact copy of an existing fp32 volume (uploaded by the caller); flag bit 0: some cost has no compact form,
// bit 1: some cost is NaN (the scan-line kernels are built NaN-free; see run_passes)
__global__ void __launch_bounds__(256) k_compact16(const float *__restrict__ C, long long n, unsigned short *__restrict__ C16, unsigned *bad8)
{
    bool bad = false;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[k] = c16_encode((i + k < n) ? C[i + k] : 0.0f);
            bad |= w[k] > 65535u;
        }
        if (i + 3 < n) *reinterpret_cast<uint2 *>(C16 + i) = make_uint2((w[0] & 65535u) | (w[1] << 16), (w[2] & 65535u) | (w[3] << 16));
        else
            for (int k = 0; k < 4 && i + k < n; k++) C16[i + k] = (unsigned short)w[k];
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) flag_once(bad8, 1u);
}
__global__ void __launch_bounds__(256) k_compact(const float *__restrict__ C, long long n, uint8_t *__restrict__ C8,
                                                 unsigned *bad8)
{
    bool bad = false, nanv = false, hopeless = false;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float x = (i + k < n) ? C[i + k] : 0.0f;
            const unsigned b = c8_encode(x);
            nanv |= x != x;
            bad |= b > 255u;
            if (b > 255u) hopeless |= c16_encode(x) > 65535u;
            w |= (b & 255u) << (8 * k);
        }
        if (i + 3 < n) *reinterpret_cast<unsigned *>(C8 + i) = w;
        else
            for (int k = 0; k < 4 && i + k < n; k++) C8[i + k] = (uint8_t)(w >> (8 * k));
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) flag_once(bad8, 1u);
    if (__builtin_amdgcn_ballot_w64(nanv) != 0ull && (threadIdx.x & 63) == 0) flag_once(bad8, 2u);
    if (__builtin_amdgcn_ballot_w64(hopeless) != 0ull && (threadIdx.x & 63) == 0) flag_once(bad8, 8u);  // ... nor in two bytes
}

// NaN scan alone, for volumes that get no compact copy (flag bit 1)
__global__ void __launch_bounds__(256) k_nanscan(const float *__restrict__ C, long long n, unsigned *flag)
{
    bool nanv = false;
    const long long n4 = n / 4, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 f = reinterpret_cast<const float4 *>(C)[i];
        nanv |= f.x != f.x || f.y != f.y || f.z != f.z || f.w != f.w;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) nanv |= C[i] != C[i];
    if (__builtin_amdgcn_ballot_w64(nanv) != 0ull && (threadIdx.x & 63) == 0) flag_once(flag, 2u);
}
hipError_t launch_nanscan(const float *C, long long n, unsigned *flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_nanscan, dim3(256 * 16), dim3(256), 0, s, C, n, flag);
    return hipGetLastError();
}

// the fp32 volume back from its compact copy (exact: every byte decodes to the float it was made from)
__global__ void __launch_bounds__(256) k_expand16(const unsigned short *__restrict__ C16, long long n, float *__restrict__ C)
{
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            const uint2 w = *reinterpret_cast<const uint2 *>(C16 + i);
            float4 f;
            f.x = c16_decode(w.x & 65535u);
            f.y = c16_decode(w.x >> 16);
            f.z = c16_decode(w.y & 65535u);
            f.w = c16_decode(w.y >> 16);
            *reinterpret_cast<float4 *>(C + i) = f;
        } else {
            for (int k = 0; k < 4 && i + k < n; k++) C[i + k] = c16_decode(C16[i + k]);
        }
    }
}
__global__ void __launch_bounds__(256) k_expand(const uint8_t *__restrict__ C8, long long n, float *__restrict__ C)
{
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            const unsigned w = *reinterpret_cast<const unsigned *>(C8 + i);
            float4 f;
            f.x = c8_decode(w & 255u);
            f.y = c8_decode((w >> 8) & 255u);
            f.z = c8_decode((w >> 16) & 255u);
            f.w = c8_decode(w >> 24);
            *reinterpret_cast<float4 *>(C + i) = f;
        } else {
            for (int k = 0; k < 4 && i + k < n; k++) C[i + k] = c8_decode(C8[i + k]);
        }
    }
}

// the fp32 volume [pix][L] from a padded compact copy [pix][LP] (the slots L..LP-1 are dropped); one wave per pixel
__global__ void __launch_bounds__(256) k_expand_padded(const uint8_t *__restrict__ C8, int cbytes, long long npix, int L, int LP, float *__restrict__ C)
{
    const int lane = threadIdx.x & 63;
    for (long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4)
        for (int o = lane; o < L; o += 64)
            C[pix * L + o] = cbytes == 2 ? c16_decode(reinterpret_cast<const unsigned short *>(C8)[pix * LP + o]) : c8_decode(C8[pix * LP + o]);
}
hipError_t launch_expand_padded(const uint8_t *C8, int cbytes, long long npix, int L, int LP, float *C, hipStream_t s)
{
    long long nb = (npix + 3) / 4;
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(k_expand_padded, dim3((unsigned)nb), dim3(256), 0, s, C8, cbytes, npix, L, LP, C);
    return hipGetLastError();
}

hipError_t launch_expand(const uint8_t *C8, int cbytes, long long n, float *C, hipStream_t s)
{
    if (cbytes == 2) hipLaunchKernelGGL(k_expand16, dim3(256 * 16), dim3(256), 0, s, reinterpret_cast<const unsigned short *>(C8), n, C);
    else hipLaunchKernelGGL(k_expand, dim3(256 * 16), dim3(256), 0, s, C8, n, C);
    return hipGetLastError();
}

// [pix][L] -> [pix][LP] (LP > L, a multiple of 64): the label slots L..LP-1 get +INF, i.e. "no such label" (dvec.cc:129).
// Writes the fp32 copy and/or the compact copy (with its "not representable" flag).  One wave per pixel.
__global__ void __launch_bounds__(256) k_pad(const float *__restrict__ C, long long npix, int L, int LP, float *__restrict__ Cp,
                                             uint8_t *__restrict__ C8p, int cbytes, unsigned *bad8)
{
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4)
        for (int o = lane; o < LP; o += 64) {
            const float x = o < L ? C[pix * L + o] : __builtin_huge_valf();
            if (Cp) Cp[pix * LP + o] = x;
            if (C8p) {
                if (cbytes == 2) {
                    const unsigned b = c16_encode(x);
                    bad |= b > 65535u;
                    reinterpret_cast<unsigned short *>(C8p)[pix * LP + o] = (unsigned short)b;
                } else {
                    const unsigned b = c8_encode(x);
                    bad |= b > 255u;
                    C8p[pix * LP + o] = (uint8_t)b;
                }
            }
        }
    if (C8p && __builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) flag_once(bad8, 1u);
}

hipError_t launch_pad(const float *C, long long npix, int L, int LP, float *Cp, uint8_t *C8p, int cbytes, unsigned *bad8, hipStream_t s)
{
    long long nb = (npix + 3) / 4;
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(k_pad, dim3((unsigned)nb), dim3(256), 0, s, C, npix, L, LP, Cp, C8p, cbytes, bad8);
    return hipGetLastError();
}

hipError_t launch_compact(const float *C, long long n, uint8_t *C8, int cbytes, unsigned *bad8, hipStream_t s)
{
    if (cbytes == 2) hipLaunchKernelGGL(k_compact16, dim3(256 * 16), dim3(256), 0, s, C, n, reinterpret_cast<unsigned short *>(C8), bad8);
    else hipLaunchKernelGGL(k_compact, dim3(256 * 16), dim3(256), 0, s, C, n, C8, bad8);
    return hipGetLastError();
}

hipError_t launch_cost(const CostParams &p, hipStream_t s)
{
    bool taken = false;
    const hipError_t e = launch_cost_fast(p, s, &taken);  // (mgm_cost_fast.hip)
    if (taken || e != hipSuccess) return e;
    const long long npix = (long long)p.nx * p.ny;
    if (p.Lreal != p.L) return hipErrorInvalidValue;  // (padded layouts: only the kernels of mgm_cost_fast.hip write them)
    hipLaunchKernelGGL(k_cost, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- K7 -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_weights(const float *__restrict__ u, int nx, int ny, int nch, float aP,
                                                 float aThresh, float *__restrict__ w)
{
    const int sx[8] = {-1, 1, 0, 0, -1, 1, 1, -1};
    const int sy[8] = {0, 0, 1, -1, -1, -1, 1, 1};
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * 8) return;
    const int o = (int)(idx / npix);
    const long long p = idx % npix;
    const int i = (int)(p % nx), j = (int)(p / nx);
    float wvalue = 1.0f;
    const int qx = i + sx[o], qy = j + sy[o];
    if (qx >= 0 && qy >= 0 && qx < nx && qy < ny) {
        float d = 0;
        for (int c = 0; c < nch; c++) {
            const float diff = u[p + c * npix] - u[(long long)qy * nx + qx + c * npix];
            d += diff * diff;
        }
        const float Delta = d / (float)nch;
        // ws(): fabs(DeltaI) < Thresh*Thresh ? aP3 : 1   (mgm_weights.h:38-41)
        wvalue = (__builtin_fabsf(Delta) < aThresh * aThresh) ? aP : 1.0f;
    }
    w[idx] = wvalue;
}

hipError_t launch_weights(const float *u, int nx, int ny, int nch, float aP, float aThresh, float *w8, hipStream_t s)
{
    const long long n = (long long)nx * ny * 8;
    hipLaunchKernelGGL(k_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u, nx, ny, nch, aP, aThresh,
                       w8);
    return hipGetLastError();
}

// "is any weight != 1.0" (mgm_core.cc:420-422)
__global__ void __launch_bounds__(256) k_any_not_one(const float *__restrict__ w, long long n, unsigned *flag)
{
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad |= (w[i] != 1.0f);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) flag_once(flag, 1u);
}

// What values do the weights take (see launch_weight_values)?  out[1] must start as 0xffffffff, the others as 0.
__global__ void __launch_bounds__(256) k_weight_values(const float *__restrict__ w, long long n, unsigned *out)
{
    bool any = false, odd = false;
    unsigned lo = 0xffffffffu, hi = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = w[i];
        if (x == 1.0f) continue;
        any = true;
        const unsigned b = __builtin_bit_cast(unsigned, x);
        if (x > 0.0f && x < __builtin_huge_valf()) {  // positive finite: the bit patterns order like the values
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
        } else
            odd = true;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
    if (any) {
        if (lo != 0xffffffffu) {
            atomicMin(out + 1, lo);
            atomicMax(out + 2, hi);
        }
    }
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull && (threadIdx.x & 63) == 0) flag_once(out + 3, 1u);
    if ((threadIdx.x & 63) == 0) flag_once(out + 0, 1u);
}
hipError_t launch_weight_values(const float *w, long long n, unsigned *out4, hipStream_t s)
{
    hipLaunchKernelGGL(k_weight_values, dim3(256 * 8), dim3(256), 0, s, w, n, out4);
    return hipGetLastError();
}
__global__ void __launch_bounds__(256) k_wsel(const float *__restrict__ w8, long long npix, unsigned *__restrict__ sel)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) m |= (w8[k * npix + p] != 1.0f ? 1u : 0u) << k;
    sel[p] = m;
}
hipError_t launch_wsel(const float *w8, long long npix, unsigned *sel, hipStream_t s)
{
    hipLaunchKernelGGL(k_wsel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, w8, npix, sel);
    return hipGetLastError();
}

// Two-valued weights, round 5: which of its two published transforms will anybody READ?  A pixel q is a neighbour of at most MGM
// pixels per pass -- p = q - d_k, k < MGM (the pass table, mgm_core.cc:463-471) -- and reader p picks E_1 or E_a by the bit of
// ITS edge to q (plane kPassToChannel[k][pass] of its selector word).  out[q] = sel[q] | need << 8, need bit 2*pass (+1) = some
// reader of q in that pass picks E_1 (E_a): the producer computes only those (image-driven weights: the edges around a pixel
// mostly agree, so most pixels need ONE min-convolution instead of two).  Readers that never update (image border) count too:
// a superset costs time, never a wrong value.
struct WneedTab {
    int d[8][4][2];
    int plane[8][4];
};
__global__ void __launch_bounds__(256) k_wneed(const unsigned *__restrict__ sel, int nx, int ny, int MGM, WneedTab T, unsigned *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)nx * ny) return;
    const int x = (int)(q % nx), y = (int)(q / nx);
    unsigned need = 0;
    for (int p = 0; p < 8; p++)
        for (int k = 0; k < MGM; k++) {
            const int px = x - T.d[p][k][0], py = y - T.d[p][k][1];
            if (px < 0 || px >= nx || py < 0 || py >= ny) continue;
            const unsigned bit = (sel[(long long)py * nx + px] >> T.plane[p][k]) & 1u;
            need |= (bit ? 2u : 1u) << (2 * p);
        }
    out[q] = (sel[q] & 0xffu) | (need << 8);
}
hipError_t launch_wneed(const unsigned *sel, int nx, int ny, int MGM, const int (*d)[4][2], const int (*plane)[4], unsigned *out, hipStream_t s)
{
    WneedTab T;
    for (int p = 0; p < 8; p++)
        for (int k = 0; k < 4; k++) {
            T.d[p][k][0] = d[p][k][0];
            T.d[p][k][1] = d[p][k][1];
            T.plane[p][k] = plane[p][k];
        }
    const long long n = (long long)nx * ny;
    hipLaunchKernelGGL(k_wneed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sel, nx, ny, MGM, T, out);
    return hipGetLastError();
}

// Placement probe (round 5): the store pattern of the pass kernels on the Lr workspace without their arithmetic -- `nstreams`
// volumes `stride` floats apart written at the same offsets at once, 16 bytes per lane.  What it is for: mgm_ctx.hip, reserve_lr.
__global__ void __launch_bounds__(256) k_probe_streams(float *base, long long stride, int nstreams, long long floats_per_stream)
{
    const long long n4 = floats_per_stream / 4;
    const float4 v = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        for (int q = 0; q < nstreams; q++) reinterpret_cast<float4 *>(base + q * stride)[i] = v;
}
hipError_t launch_probe_streams(float *base, long long stride, int nstreams, long long floats_per_stream, hipStream_t s)
{
    hipLaunchKernelGGL(k_probe_streams, dim3(256 * 16), dim3(256), 0, s, base, stride, nstreams, floats_per_stream);
    return hipGetLastError();
}

// Debug check of the self-validating hand-off slabs (mgm_pass2.hip, TAGS): after a launch EVERY word of the slots its
// passes own must carry the launch's tag in its sign bit -- the invariant the protocol rests on ("each slot is written
// exactly once per launch of its pass").  Counts the words that do not.
__global__ void __launch_bounds__(256) k_check_tags(const unsigned *__restrict__ w, long long n, unsigned tag, unsigned *count)
{
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad += ((w[i] ^ tag) >> 31);
    if (bad) atomicAdd(count, bad);
}
// which XCC ids the workgroups of a launch see (bit i: some workgroup ran on XCD i)
__global__ void k_xcc_census(unsigned *mask)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicOr(mask, 1u << (xcc & 31u));
}
hipError_t launch_xcc_census(unsigned *mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_xcc_census, dim3(2048), dim3(64), 0, s, mask);
    return hipGetLastError();
}

hipError_t launch_check_tags(const float *slabs, long long nwords, unsigned tag, unsigned *count, hipStream_t s)
{
    long long blocks = (nwords + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_check_tags, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const unsigned *>(slabs), nwords, tag, count);
    return hipGetLastError();
}

hipError_t launch_any_not_one(const float *w, long long n, unsigned *flag, hipStream_t s)
{
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_any_not_one, dim3((unsigned)blocks), dim3(256), 0, s, w, n, flag);
    return hipGetLastError();
}

}  // namespace mgm
