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
#include "mgm_device.h"

namespace mgm {

__device__ __forceinline__ bool finite_bits(float x)
{
    return (__builtin_bit_cast(unsigned, x) & 0x7f800000u) != 0x7f800000u;
}

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

// ---- the costs that look at more than one sample per image ------------------------
// Birchfield-Tomasi dissimilarity of one channel (mgm_costvolume.h:82-110).  Each sample spans the closed interval
// between itself and its two half-way interpolants along x (at the image border the interpolant is the sample; the
// halving is a double operation narrowed back to float, as compiled there); the dissimilarity is the smaller of
// the two one-sided distances "sample of one image to the interval of the other".  The three-way selections keep
// the reference's comparison tree, which decides what a NaN sample does.
__device__ __forceinline__ float tri_low(float x, float y, float z)
{
    if (x < y) return x < z ? x : z;
    return z < y ? z : y;
}
__device__ __forceinline__ float tri_high(float x, float y, float z)
{
    if (x > y) return x > z ? x : z;
    return z > y ? z : y;
}
struct BtSpan {
    float centre, lo, hi;
};
__device__ __forceinline__ BtSpan bt_span(const float *__restrict__ row, int width, int x)
{
    const float c = row[x];
    float ahead = c, behind = c;
    if (x + 1 < width) ahead = (float)((double)(c + row[x + 1]) * 0.5);
    if (x > 0) behind = (float)((double)(c + row[x - 1]) * 0.5);
    return BtSpan{c, tri_low(behind, ahead, c), tri_high(behind, ahead, c)};
}
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

// Raise a bit of a flag word that many waves may want to raise: look first -- an atomic per wave on ONE address serialises
// (round 4: RGB absolute differences exceed 254 at nearly every pixel, and 2 M atomicOr on the "no compact form" word made
// K2 take 21.8 ms at 1920x1080x256 where the grey-level volume took 1.55).
__device__ __forceinline__ void flag_once(unsigned *word, unsigned bit)
{
    if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u) atomicOr(word, bit);
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

// K2 for the case whose costs are known to fit the compact form (single-word census, trunc = +INF or an
// integer <= 254; see mgm_costvolume_build_dev): integer arithmetic only, the compact volume only.
// One wavefront per pixel; lane l owns the LPL consecutive labels l*LPL.. -- LPL bytes, one store.
//   cost = min(popcount(cu ^ cv), trunc), trunc for a hypothesis outside the right image
//   (mgm_costvolume.h:65-78, 401-412); a pixel without a finite cost is all zeros (414-421).
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int LPL>
__global__ void __launch_bounds__(256) k_cost_census8(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                      int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                      uint8_t *__restrict__ C8, int Lreal)
{
    constexpr int L = LPL * 64;
    const long long npix = (long long)nx * ny;
    const int lane = threadIdx.x & 63;
    for (long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4) {
        const int x = (int)(pix % nx), y = (int)(pix / nx);
        const uint32_t wu = cu[pix];
        const int q0 = x + dmin + lane * LPL;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned b[LPL];
        if (yin && q0 >= 0 && q0 + LPL <= vnx && (lane + 1) * LPL <= Lreal) {  // the whole group lies inside the right image
            uint32_t wv[LPL];
            if constexpr (LPL % 4 == 0) {
#pragma unroll
                for (int h = 0; h < LPL / 4; h++) {
                    const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
                    wv[4 * h] = t.x; wv[4 * h + 1] = t.y; wv[4 * h + 2] = t.z; wv[4 * h + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < LPL; k++) wv[k] = row[q0 + k];
            }
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const unsigned pc = (unsigned)__builtin_popcount(wu ^ wv[k]);
                b[k] = pc < tb ? pc : tb;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LPL; k++) {
                const int q = q0 + k;
                const bool in = yin && q >= 0 && q < vnx;
                const unsigned pc = (unsigned)__builtin_popcount(wu ^ row[in ? q : 0]);
                b[k] = in ? (pc < tb ? pc : tb) : tb;
                if (lane * LPL + k >= Lreal) b[k] = 255u;  // a slot of the padded layout: +INF
            }
        }
        bool fin = false;
#pragma unroll
        for (int k = 0; k < LPL; k++) fin |= b[k] != 255u;
        if (__builtin_amdgcn_ballot_w64(fin) == 0ull) {  // no valid hypothesis: zeros (the slots of a padded layout stay +INF)
#pragma unroll
            for (int k = 0; k < LPL; k++) b[k] = lane * LPL + k >= Lreal ? 255u : 0u;
        }
        uint8_t *dst = C8 + pix * L + lane * LPL;
        if constexpr (LPL == 1) {
            dst[0] = (uint8_t)b[0];
        } else if constexpr (LPL == 2) {
            *reinterpret_cast<unsigned short *>(dst) = (unsigned short)(b[0] | (b[1] << 8));
        } else if constexpr (LPL % 4 == 0) {
#pragma unroll
            for (int h = 0; h < LPL / 4; h++)
                reinterpret_cast<unsigned *>(dst)[h] = b[4 * h] | (b[4 * h + 1] << 8) | (b[4 * h + 2] << 16) | (b[4 * h + 3] << 24);
        } else {
#pragma unroll
            for (int k = 0; k < LPL; k++) dst[k] = (uint8_t)b[k];
        }
    }
}

// The same costs for label counts that divide 1024, sixteen labels per lane: a wave writes 1 KiB = 1024 / L whole
// pixels per iteration with 16-byte stores (the 4-byte version above spends its time in per-pixel index arithmetic
// and load latency: one pixel per wave and iteration).
template <int L>
__global__ void __launch_bounds__(256) k_cost_census8w(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                       int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                       uint8_t *__restrict__ C8)
{
    static_assert(L == 64 || L == 128 || L == 256 || L == 512, "whole pixels per KiB");
    constexpr int LP = L / 16;    // lanes per pixel
    constexpr int PPC = 64 / LP;  // pixels per wave and iteration
    const long long npix = (long long)nx * ny;
    const long long nchunk = (npix + PPC - 1) / PPC;
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane % LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP % 64)) - 1ull)) << (sub * LP);
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long pix = chunk * PPC + sub;
        const bool live = pix < npix;
        const unsigned p32 = live ? (unsigned)pix : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(p32 / (unsigned)nx), x = (int)(p32 - (unsigned)y * (unsigned)nx);
        const uint32_t wu = cu[p32];
        const int q0 = x + dmin + part * 16;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned w[4];
        bool fin = false;
        if (yin && q0 >= 0 && q0 + 16 <= vnx) {  // the lane's sixteen labels lie inside the right image
            u32x4_a4 t[4];
#pragma unroll
            for (int h = 0; h < 4; h++) t[h] = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const unsigned v[4] = {t[h].x, t[h].y, t[h].z, t[h].w};
                unsigned b[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned pc = (unsigned)__builtin_popcount(wu ^ v[k]);
                    b[k] = pc < tb ? pc : tb;
                    fin |= b[k] != 255u;
                }
                w[h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; h++) {
                unsigned b[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q = q0 + 4 * h + k;
                    const bool in = yin && q >= 0 && q < vnx;
                    const unsigned pc = (unsigned)__builtin_popcount(wu ^ row[in ? q : 0]);
                    b[k] = in ? (pc < tb ? pc : tb) : tb;
                    fin |= b[k] != 255u;
                }
                w[h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            }
        }
        const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin) & group) != 0ull;  // of this pixel's labels
        if (live) {
            uint4 o;
            o.x = anyfinite ? w[0] : 0u; o.y = anyfinite ? w[1] : 0u; o.z = anyfinite ? w[2] : 0u; o.w = anyfinite ? w[3] : 0u;
            *reinterpret_cast<uint4 *>(C8 + pix * L + part * 16) = o;
        }
    }
}

// The same again with FOUR consecutive pixels of a row per lane (image widths that are multiples of four; any compact
// label count -- at 192 / 384 labels a pixel group takes 12 / 24 lanes and the last 4 / 16 lanes of the wave idle): the sixteen
// labels of a lane slide along the right image by one word per pixel, so the four pixels share 19 census words where
// four separate lanes load 64 -- the kernel above is bound by those (L1-resident, unaligned) loads, not by its stores.
template <int L>
__global__ void __launch_bounds__(256) k_cost_census8x(const uint32_t *__restrict__ cu, const uint32_t *__restrict__ cv,
                                                       int nx, int ny, int vnx, int vny, int dmin, unsigned tb,
                                                       uint8_t *__restrict__ C8, int Lreal)
{
    static_assert(L % 16 == 0 && L >= 16 && L <= 1024, "sixteen labels per lane");
    constexpr int LP = L / 16;    // lanes per pixel group
    constexpr int G = 64 / LP;    // groups of four pixels per wave and iteration (192 / 384 labels: 4 / 16 lanes of the wave idle)
    const long long npix = (long long)nx * ny;  // (a multiple of four)
    const long long nchunk = (npix + 4 * G - 1) / (4 * G);
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane % LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP % 64)) - 1ull)) << ((sub * LP) & 63);
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long pix0 = (chunk * G + sub) * 4;
        const bool live = sub < G && pix0 < npix;
        const unsigned p32 = live ? (unsigned)pix0 : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(p32 / (unsigned)nx), x = (int)(p32 - (unsigned)y * (unsigned)nx);  // x .. x+3: one row
        const uint4 wu4 = *reinterpret_cast<const uint4 *>(cu + p32);
        const unsigned wu[4] = {wu4.x, wu4.y, wu4.z, wu4.w};
        const int q0 = x + dmin + part * 16;
        const bool yin = y < vny;
        const uint32_t *row = cv + (long long)(yin ? y : 0) * vnx;
        unsigned w[4][4];
        bool fin[4] = {false, false, false, false};
        unsigned pad[4];  // what a pixel without a valid hypothesis gets: zeros, the slots of a padded layout +INF
#pragma unroll
        for (int h = 0; h < 4; h++) {
            pad[h] = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) pad[h] |= (part * 16 + 4 * h + k >= Lreal ? 255u : 0u) << (8 * k);
        }
        if (yin && q0 >= 0 && q0 + 20 <= vnx && (part + 1) * 16 <= Lreal) {  // every word the four pixels need lies inside the right image
            unsigned v[20];
#pragma unroll
            for (int h = 0; h < 5; h++) {
                const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(row + q0 + 4 * h);
                v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
            }
            // (a bit count is at most 32: never the +INF code, and clipped only by a truncation below 32 -- wave-uniform)
            const bool clip = tb < 32u;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                fin[i] = true;
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    unsigned b[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) b[k] = (unsigned)__builtin_popcount(wu[i] ^ v[i + 4 * h + k]);
                    if (clip) {
#pragma unroll
                        for (int k = 0; k < 4; k++) b[k] = b[k] < tb ? b[k] : tb;
                    }
                    w[i][h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    unsigned b[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int q = q0 + i + 4 * h + k;
                        const bool in = yin && q >= 0 && q < vnx;
                        const unsigned pc = (unsigned)__builtin_popcount(wu[i] ^ row[in ? q : 0]);
                        b[k] = in ? (pc < tb ? pc : tb) : tb;
                        if (part * 16 + 4 * h + k >= Lreal) b[k] = 255u;  // a slot of the padded layout: +INF
                        fin[i] |= b[k] != 255u;
                    }
                    w[i][h] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin[i]) & group) != 0ull;  // of this pixel's labels
            if (live) {
                uint4 o;
                o.x = anyfinite ? w[i][0] : pad[0]; o.y = anyfinite ? w[i][1] : pad[1]; o.z = anyfinite ? w[i][2] : pad[2]; o.w = anyfinite ? w[i][3] : pad[3];
                *reinterpret_cast<uint4 *>(C8 + (pix0 + i) * L + part * 16) = o;
            }
        }
    }
}

// ---- absolute / squared differences, compact form only (round 4) ---------------------------------------------------
// computeC_AD / computeC_SD (mgm_costvolume.h:23-44) for a volume that is EXPECTED to fit the compact form (8-bit images:
// whole-number differences): only the compact copy is written, CB bytes per cost, and the flag word says afterwards whether
// every cost really had that form -- if one did not, mgm_costvolume_build_dev runs the general kernel, which writes the fp32
// volume.  Work layout of k_cost_census8x: four consecutive pixels of a row and sixteen labels per lane, the four pixels
// share NL + 3 samples of the right image per channel; one 16-byte store per lane and pixel (NL = 16 or 8 labels).
//   * The sum over the channels runs in channel order from 0, as there.  x = max(d, -d) enters as |d| (a source modifier):
//     the two differ in the sign of a zero or of a NaN only, and 0 + x, x * x and "NaN loses the comparison with truncDist"
//     hide both.
//   * truncDist is +INF or a non-negative number here (the caller checks), so min(e, truncDist) is v_min_f32: a NaN cost
//     becomes truncDist exactly as with the reference's comparison.
//   * Encoding: convert, and keep the largest cost and the largest fractional part of the lane's costs of a pixel -- only a
//     lane that saw a cost outside [0, LIM] or a fraction (labels
//     outside the right image with truncDist = +INF; volumes that will be filled again) takes the careful c8_encode /
//     c16_encode path.
// NCH = the channel count (1 or 3: the right-image samples of all channels are loaded up front), or 0 = any (channel loop
// outermost, 64 accumulators).
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// four consecutive samples starting at p (a group of four pixels of one row; `n` < 4 of them exist at the end of a row whose
// width is not a multiple of four: the others repeat the first)
__device__ __forceinline__ void load4_row(const float *__restrict__ p, int n, float (&o)[4])
{
    if (n >= 4) {
        const f32x4_a4 w = *reinterpret_cast<const f32x4_a4 *>(p);
        o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = p[i < n ? i : 0];
    }
}
template <int NL>
__device__ __forceinline__ void diff_load(const CostParams &P, int t, long long pix0, int nlive, long long npix, long long vpix, int y, bool yin, int q0,
                                          bool inside, float (&ut)[4], float (&vt)[NL + 4])
{
    load4_row(P.u + (long long)t * npix + pix0, nlive, ut);
    const float *row = P.v + (long long)t * vpix + (long long)(yin ? y : 0) * P.vnx;
    if (inside) {
#pragma unroll
        for (int h = 0; h < NL / 4 + 1; h++) {
            const f32x4_a4 w = *reinterpret_cast<const f32x4_a4 *>(row + q0 + 4 * h);
            vt[4 * h] = w.x; vt[4 * h + 1] = w.y; vt[4 * h + 2] = w.z; vt[4 * h + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NL + 3; k++) {
            const int q = q0 + k;
            vt[k] = row[(yin && q >= 0 && q < P.vnx) ? q : 0];
        }
        vt[NL + 3] = 0.0f;
    }
}
template <int CB, int NCH, bool SD>
__global__ void __launch_bounds__(256) k_cost_diffx(const CostParams P)
{
    constexpr unsigned LIM = CB == 2 ? 65534u : 254u;  // the largest finite code
    constexpr int NC = NCH ? NCH : 1;
    constexpr int NL = 16 / CB;      // labels per lane: sixteen bytes, one store per pixel
    const int L = P.L, LP = L / NL;  // lanes per group of four pixels
    const int G = 64 / LP;           // groups per wave and iteration (192 / 384 / 768 labels: the last lanes of the wave idle)
    const int nx = P.nx, vnx = P.vnx;
    const long long npix = (long long)nx * P.ny, vpix = (long long)vnx * P.vny;
    const int gpr = (nx + 3) / 4;  // groups of four pixels per row (the last one of a row may hold fewer)
    const long long ngrp = (long long)gpr * P.ny, nchunk = (ngrp + G - 1) / G;
    const int lane = threadIdx.x & 63, sub = lane / LP, part = lane - sub * LP;
    const unsigned long long group = (LP == 64 ? ~0ull : ((1ull << (LP & 63)) - 1ull)) << ((sub * LP) & 63);
    const float trunc = P.trunc, tclamp = __builtin_fminf(trunc, (float)(LIM + 2u));
    const bool padlane = (part + 1) * NL > P.Lreal;  // this lane holds label slots of a padded layout (P.Lreal < P.L)
    bool odd = false, hopeless = CB == 2;  // a cost without the compact form of this width / of either width
    for (long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunk; chunk += (long long)gridDim.x * 4) {
        const long long grp = chunk * G + sub;
        const bool live = sub < G && grp < ngrp;
        const unsigned g32 = live ? (unsigned)grp : 0u;  // (npix < 2^31: checked by the caller)
        const int y = (int)(g32 / (unsigned)gpr), x = (int)(g32 - (unsigned)y * (unsigned)gpr) * 4;  // x .. x+3: one row
        const long long pix0 = (long long)y * nx + x;
        const int nlive = live ? (nx - x < 4 ? nx - x : 4) : 0, nload = nx - x < 4 ? nx - x : 4;
        const int q0 = x + P.dmin + part * NL;
        const bool yin = y < P.vny;
        const bool inside = yin && q0 >= 0 && q0 + NL + 4 <= vnx;  // every sample the four pixels need lies inside the right image
        float uu[NC][4], v[NC][NL + 4];
        float e[NCH ? 1 : 4][NL];
        if constexpr (NCH != 0) {
#pragma unroll
            for (int t = 0; t < NCH; t++) diff_load<NL>(P, t, pix0, nload, npix, vpix, y, yin, q0, inside, uu[t], v[t]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < NL; k++) e[i][k] = 0.0f;
            for (int t = 0; t < P.nch; t++) {
                diff_load<NL>(P, t, pix0, nload, npix, vpix, y, yin, q0, inside, uu[0], v[0]);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < NL; k++) {
                        const float d = uu[0][i] - v[0][i + k];
                        e[i][k] += SD ? d * d : __builtin_fabsf(d);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float c[NL];
            unsigned b[NL];
            float top = 0.0f, frac = 0.0f;  // the largest clamped cost and the largest fractional part
#pragma unroll
            for (int k = 0; k < NL; k++) {
                if constexpr (NCH != 0) {
                    float a = 0.0f;  // (0 + x: the compiler drops it; x >= +0)
#pragma unroll
                    for (int t = 0; t < NCH; t++) {
                        const float d = uu[t][i] - v[t][i + k];
                        a += SD ? d * d : __builtin_fabsf(d);
                    }
                    c[k] = a;
                } else {
                    c[k] = e[i][k];
                }
                if (!inside) {  // a label outside the right image costs truncDist (mgm_costvolume.h:401-412)
                    const int q = q0 + i + k;
                    c[k] = (yin && q >= 0 && q < vnx) ? c[k] : trunc;
                }
                float cc = __builtin_fminf(c[k], tclamp);  // in [0, LIM + 2]: the conversion is defined
                if (padlane && part * NL + k >= P.Lreal) cc = (float)(LIM + 1u);  // a slot of the padded layout: +INF, as its code
                b[k] = (unsigned)cc;
                top = __builtin_fmaxf(top, cc);
                frac = __builtin_fmaxf(frac, __builtin_amdgcn_fractf(cc));
            }
            bool fin = true;
            if (frac > 0.0f || top > (float)LIM) {  // the careful path
                fin = false;
#pragma unroll
                for (int k = 0; k < NL; k++) {
                    const float ct = (padlane && part * NL + k >= P.Lreal) ? __builtin_huge_valf() : ((c[k] < trunc) ? c[k] : trunc);
                    fin |= finite_bits(ct);
                    b[k] = CB == 2 ? c16_encode(ct) : c8_encode(ct);
                    odd |= b[k] > LIM + 1u && i < nlive;  // (i >= nlive: not a pixel of the image, see load4_row)
                    if (CB == 1) hopeless |= c16_encode(ct) > 65535u;  // (... nor in two bytes)
                }
            }
            // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
            const bool anyfinite = (__builtin_amdgcn_ballot_w64(fin) & group) != 0ull;
            if (i >= nlive) continue;
            uint8_t *dst = P.C8 + ((pix0 + i) * L + part * NL) * CB;
            if constexpr (CB == 2) {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = anyfinite ? ((b[2 * k] & 65535u) | (b[2 * k + 1] << 16))
                                     : ((padlane && part * NL + 2 * k >= P.Lreal ? 65535u : 0u) | (padlane && part * NL + 2 * k + 1 >= P.Lreal ? 65535u << 16 : 0u));
                reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    w[k] = anyfinite ? ((b[4 * k] & 255u) | ((b[4 * k + 1] & 255u) << 8) | ((b[4 * k + 2] & 255u) << 16) | (b[4 * k + 3] << 24))
                                     : ((padlane && part * NL + 4 * k >= P.Lreal ? 255u : 0u) | (padlane && part * NL + 4 * k + 1 >= P.Lreal ? 255u << 8 : 0u) |
                                        (padlane && part * NL + 4 * k + 2 >= P.Lreal ? 255u << 16 : 0u) | (padlane && part * NL + 4 * k + 3 >= P.Lreal ? 255u << 24 : 0u));
                reinterpret_cast<uint4 *>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    // flag bit 0: some cost has no compact form of this width; bit 3: ... and a wider one would not help either
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull && lane == 0) flag_once(P.bad8, 1u);
    if (__builtin_amdgcn_ballot_w64(odd && hopeless) != 0ull && lane == 0) flag_once(P.bad8, 8u);
}
template <int CB, bool SD>
static void launch_diffx(const CostParams &p, hipStream_t s)
{
    const long long npix = (long long)p.nx * p.ny;
    long long nw = ((long long)((p.nx + 3) / 4) * p.ny * 4 * p.L * CB / 4096 + 3) / 4 + 1;
    if (nw > 256 * 32) nw = 256 * 32;
    const dim3 grid((unsigned)nw), block(256);
    if (p.nch == 1) hipLaunchKernelGGL((k_cost_diffx<CB, 1, SD>), grid, block, 0, s, p);
    else if (p.nch == 3) hipLaunchKernelGGL((k_cost_diffx<CB, 3, SD>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((k_cost_diffx<CB, 0, SD>), grid, block, 0, s, p);
}

// ---- Birchfield-Tomasi costs, restructured (round 4) --------------------------------------------------------------------
// computeC_BTAD / computeC_BTSD (mgm_costvolume.h:82-135) look at three samples of each image per cell -- but the interval a
// sample spans depends on its own image alone: k_bt_spans writes the two ends once per sample (2*nch planes per image, same
// operations as bt_span above), and k_cost_btx is left with two three-way maxima and a minimum per cell and channel.  Work
// layout of k_cost_diffx: a wave takes four consecutive pixels of a row, a lane four consecutive labels of them (and the
// next 256 labels in its next turn): one 16-byte store of fp32 costs per lane and pixel -- the costs are multiples of one
// half, there is no compact form for them.
__global__ void __launch_bounds__(256) k_bt_spans(const float *__restrict__ u, int nx, int ny, int nch, float *__restrict__ sp)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix * nch) return;
    const int t = (int)(idx / npix);
    const long long p = idx - t * npix;
    const int y = (int)(p / nx), x = (int)(p - (long long)y * nx);
    const BtSpan a = bt_span(u + t * npix + (long long)y * nx, nx, x);
    sp[idx] = a.lo;
    sp[idx + npix * nch] = a.hi;
}
// FN = the cost function (CostParams::costfn): 4 / 5 Birchfield-Tomasi as described; 0 / 1 absolute / squared differences and 2
// census over several descriptor words, for the volumes of those that have no compact form (float-valued or blurred
// images, costs that are thirds or halves of bit counts) and used to take the general kernel: the same layout, fp32 out.
template <int FN, bool W4>  // W4: the image width is a multiple of four (every group is whole: no guarded loads and stores)
__global__ void __launch_bounds__(256) k_cost_btx(const CostParams P)
{
    constexpr bool BT = FN >= 4, SD = FN == 5 || FN == 1;
    const int nx = P.nx, vnx = P.vnx, L = P.L, nch = P.nch;
    const long long npix = (long long)nx * P.ny, vpix = (long long)vnx * P.vny;
    const int gpr = (nx + 3) / 4;  // groups of four pixels per row (the last one of a row may hold fewer)
    const long long ngroup = (long long)gpr * P.ny;
    const int lane = threadIdx.x & 63;
    const float trunc = P.trunc;
    bool nanv = false;
    for (long long grp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngroup; grp += (long long)gridDim.x * 4) {
        const int y = (int)(grp / gpr), x = (int)(grp - (long long)y * gpr) * 4;  // x .. x+3: one row
        const long long pix0 = (long long)y * nx + x;
        const int nlive = W4 ? 4 : (nx - x < 4 ? nx - x : 4);
        const bool yin = y < P.vny;
        bool fin[4] = {false, false, false, false};
        for (int o0 = lane * 4; o0 < L; o0 += 256) {
            const int q0 = x + P.dmin + o0;
            const bool inside = yin && q0 >= 0 && q0 + 8 <= vnx;  // every sample the four pixels need lies inside the right image
            float e[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 4; k++) e[i][k] = 0.0f;
            for (int t = 0; t < nch; t++) {
                float ac[4], al[4] = {}, ah[4] = {};
                load4_row(P.u + (long long)t * npix + pix0, nlive, ac);
                if constexpr (BT) {
                    load4_row(P.ncc_u + (long long)t * npix + pix0, nlive, al);
                    load4_row(P.ncc_u + (long long)(nch + t) * npix + pix0, nlive, ah);
                }
                const long long rowoff = (long long)(yin ? y : 0) * vnx;
                const float *rc = P.v + (long long)t * vpix + rowoff;
                const float *rl = BT ? P.ncc_v + (long long)t * vpix + rowoff : rc, *rh = BT ? P.ncc_v + (long long)(nch + t) * vpix + rowoff : rc;
                float bc[8], bl[8] = {}, bh[8] = {};
                if (inside) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const f32x4_a4 c4 = *reinterpret_cast<const f32x4_a4 *>(rc + q0 + 4 * h);
                        bc[4 * h] = c4.x; bc[4 * h + 1] = c4.y; bc[4 * h + 2] = c4.z; bc[4 * h + 3] = c4.w;
                        if constexpr (BT) {
                            const f32x4_a4 l4 = *reinterpret_cast<const f32x4_a4 *>(rl + q0 + 4 * h);
                            const f32x4_a4 h4 = *reinterpret_cast<const f32x4_a4 *>(rh + q0 + 4 * h);
                            bl[4 * h] = l4.x; bl[4 * h + 1] = l4.y; bl[4 * h + 2] = l4.z; bl[4 * h + 3] = l4.w;
                            bh[4 * h] = h4.x; bh[4 * h + 1] = h4.y; bh[4 * h + 2] = h4.z; bh[4 * h + 3] = h4.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        const int q = q0 + k;
                        const int qq = (yin && q >= 0 && q < vnx) ? q : 0;
                        bc[k] = rc[qq];
                        if constexpr (BT) {
                            bl[k] = rl[qq];
                            bh[k] = rh[qq];
                        }
                    }
                    bc[7] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if constexpr (BT) {
                            const float a_to_b = tri_high(0.0f, ac[i] - bh[i + k], bl[i + k] - ac[i]);
                            const float b_to_a = tri_high(0.0f, bc[i + k] - ah[i], al[i] - bc[i + k]);
                            const float r = __builtin_fabsf(a_to_b < b_to_a ? a_to_b : b_to_a);
                            e[i][k] += SD ? r * r : r;
                        } else if constexpr (FN == 2) {  // the samples are descriptor words (mgm_costvolume.h:65-78)
                            e[i][k] += (float)__builtin_popcount(__builtin_bit_cast(unsigned, ac[i]) ^ __builtin_bit_cast(unsigned, bc[i + k]));
                        } else {  // computeC_AD / computeC_SD (23-44)
                            float d = ac[i] - bc[i + k];
                            d = (d > -d) ? d : -d;
                            e[i][k] += SD ? d * d : d;
                        }
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float c[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q = q0 + i + k;
                    float v = e[i][k];
                    if constexpr (FN == 2) v = (float)((double)v * 1.0 / (double)nch);
                    c[k] = (inside || (yin && q >= 0 && q < vnx)) ? v : trunc;  // outside the right image: truncDist (401-412)
                    c[k] = (c[k] < trunc) ? c[k] : trunc;
                    fin[i] |= finite_bits(c[k]);
                    nanv |= c[k] != c[k];
                }
                if (i < nlive) *reinterpret_cast<float4 *>(P.C + (pix0 + i) * L + o0) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
        // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (__builtin_amdgcn_ballot_w64(fin[i]) == 0ull && i < nlive)
                for (int o0 = lane * 4; o0 < L; o0 += 256) *reinterpret_cast<float4 *>(P.C + (pix0 + i) * L + o0) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (P.bad8 && __builtin_amdgcn_ballot_w64(nanv) != 0ull && lane == 0) flag_once(P.bad8, 2u);
}

// ---- clipped NCC, restructured (round 4) ---------------------------------------------------------------------------
// computeC_clippedNCC (mgm_costvolume.h:137-165) accumulates five window sums per (pixel, label, channel) -- but mu1 and s1
// depend on the left pixel alone and mu2, s2 on the right pixel alone: only the cross term is per cell.  Each sum is a
// sequential fp32 accumulation over the window in the reference's (i outer, j inner) order, so computing it ONCE per pixel
// in that order gives the same bits as computing it per label; likewise s - mu*mu (one rounded product, one rounded
// difference).  k_ncc_stats does that for both images (and notes whether the window lies inside the image and is NaN-free:
// otherwise the reference returns INFINITY whatever the other window holds); k_cost_ncc then needs 25 products per cell
// instead of 125 operations and 50 loads, with the rows of both images staged in LDS.
__global__ void __launch_bounds__(256) k_ncc_stats(const float *__restrict__ u, int nx, int ny, int nch, int hw, float *__restrict__ st)
{
    const long long npix = (long long)nx * ny;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const int x = (int)(idx % nx), y = (int)(idx / nx);
    bool ok = x - hw >= 0 && y - hw >= 0 && x + hw < nx && y + hw < ny;
    for (int t = 0; t < nch; t++) {
        float mu = 0, s2 = 0;
        int n = 0;
        if (ok)
            for (int i = -hw; i <= hw; i++)
                for (int j = -hw; j <= hw; j++) {
                    const float v = u[(x + i) + (long long)(y + j) * nx + t * npix];
                    ok = ok && (v == v);
                    mu += v;
                    s2 += v * v;
                    n++;
                }
        n = n ? n : 1;
        mu /= n;
        s2 /= n;
        st[idx + (long long)t * npix] = mu;
        st[idx + (long long)(nch + t) * npix] = s2 - mu * mu;
    }
    st[idx + (long long)(2 * nch) * npix] = ok ? 1.0f : 0.0f;
}

// One wavefront per pixel, lane l takes the labels l, l+64, ...; a workgroup of four waves walks PXB consecutive pixels of
// one image row with the 2*hw+1 rows of both images around it in LDS (conflict-free: consecutive lanes read consecutive
// words; the left window is a broadcast read).
constexpr int kNccPxb = 32;       // pixels of a row per workgroup
constexpr int kNccMaxHw = 3;      // windows up to 7x7 (CENSUS_NCC_WIN <= 7); wider ones take the general kernel
constexpr int kNccMaxL = 1024;    // LDS: (PXB + L + 2*hw) floats per row and channel
template <int HW>
__global__ void __launch_bounds__(256) k_cost_ncc(const CostParams P)
{
    constexpr int WIN = 2 * HW + 1;
    extern __shared__ float ncc_lds[];
    const int nch = P.nch, L = P.L;
    const int ntx = (P.nx + kNccPxb - 1) / kNccPxb;
    const int y = blockIdx.x / ntx, x0 = (blockIdx.x % ntx) * kNccPxb;
    const int uw = kNccPxb + 2 * HW;           // staged columns of the left image: x0-HW ..
    const int vw = kNccPxb + L - 1 + 2 * HW;   // ... of the right image: x0+dmin-HW ..
    float *Lu = ncc_lds;                       // [nch][WIN][uw]
    float *Lv = Lu + nch * WIN * uw;           // [nch][WIN][vw]
    const long long npix = (long long)P.nx * P.ny, vpix = (long long)P.vnx * P.vny;
    for (int k = threadIdx.x; k < nch * WIN * uw; k += blockDim.x) {
        const int c = k % uw, r = (k / uw) % WIN, t = k / (uw * WIN);
        const int xx = x0 - HW + c, yy = y - HW + r;
        Lu[k] = (xx >= 0 && xx < P.nx && yy >= 0 && yy < P.ny) ? P.u[xx + (long long)yy * P.nx + t * npix] : 0.0f;
    }
    for (int k = threadIdx.x; k < nch * WIN * vw; k += blockDim.x) {
        const int c = k % vw, r = (k / vw) % WIN, t = k / (vw * WIN);
        const int xx = x0 + P.dmin - HW + c, yy = y - HW + r;
        Lv[k] = (xx >= 0 && xx < P.vnx && yy >= 0 && yy < P.vny) ? P.v[xx + (long long)yy * P.vnx + t * vpix] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool yin = y < P.vny;
    for (int xl = wave; xl < kNccPxb; xl += 4) {
        const int x = x0 + xl;
        if (x >= P.nx) break;
        const long long pix = (long long)y * P.nx + x;
        float *Cp = P.C + pix * L;
        const bool ok1 = P.ncc_u[pix + (long long)(2 * nch) * npix] != 0.0f;
        bool anyfinite = false, nanv = false;
        for (int o = lane; o < L; o += 64) {
            const int qx = x + o + P.dmin;
            float e = P.trunc;
            if (yin && qx >= 0 && qx < P.vnx) {
                const long long q = (long long)y * P.vnx + qx;
                if (!ok1 || P.ncc_v[q + (long long)(2 * nch) * vpix] == 0.0f) {
                    e = __builtin_huge_valf();
                } else {
                    float NCC = 0;
                    for (int t = 0; t < nch; t++) {
                        const float *a = Lu + (t * WIN) * uw + xl;            // left window: column xl + (i + HW), row j + HW
                        const float *b = Lv + (t * WIN) * vw + xl + o;        // right window: column xl + o + (i + HW)
                        float prod = 0;
#pragma unroll
                        for (int i = 0; i < WIN; i++)
#pragma unroll
                            for (int j = 0; j < WIN; j++) prod += a[j * uw + i] * b[j * vw + i];
                        prod /= (WIN * WIN);
                        const float mu1 = P.ncc_u[pix + (long long)t * npix], mu2 = P.ncc_v[q + (long long)t * vpix];
                        const float var = P.ncc_u[pix + (long long)(nch + t) * npix] * P.ncc_v[q + (long long)(nch + t) * vpix];
                        const double den = (0.0000001 > var) ? 0.0000001 : (double)var;
                        NCC = (float)(NCC + (prod - mu1 * mu2) / __builtin_sqrt(den));
                    }
                    const float m = (NCC < nch) ? NCC : (float)nch;
                    const float c = (0 > m) ? 0 : m;
                    const float clipped = nch - c;
                    e = clipped * 64;
                }
            }
            e = (e < P.trunc) ? e : P.trunc;
            Cp[o] = e;
            anyfinite |= finite_bits(e);
            nanv |= e != e;
        }
        // no valid hypothesis for this pixel => all labels cost 0 (mgm_costvolume.h:414-421)
        if (__builtin_amdgcn_ballot_w64(anyfinite) == 0ull)
            for (int o = lane; o < L; o += 64) Cp[o] = 0.0f;
        if (P.bad8 && __builtin_amdgcn_ballot_w64(nanv) != 0ull && lane == 0) flag_once(P.bad8, 2u);
    }
}

template <int FN>
static void launch_btx(const CostParams &p, long long nw, hipStream_t s)
{
    if (p.nx % 4) hipLaunchKernelGGL((k_cost_btx<FN, false>), dim3((unsigned)nw), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_cost_btx<FN, true>), dim3((unsigned)nw), dim3(256), 0, s, p);
}
hipError_t launch_cost(const CostParams &p, hipStream_t s)
{
    const long long npix = (long long)p.nx * p.ny;
    if (p.costfn == 3 && p.ncc_u && p.ncc_v && p.C && !p.C8 && !p.rlo && p.hwin >= 1 && p.hwin <= kNccMaxHw && p.L <= kNccMaxL && p.nch <= 4) {
        const long long vpix = (long long)p.vnx * p.vny;
        hipLaunchKernelGGL(k_ncc_stats, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, p.u, p.nx, p.ny, p.nch, p.hwin, p.ncc_u);
        hipLaunchKernelGGL(k_ncc_stats, dim3((unsigned)((vpix + 255) / 256)), dim3(256), 0, s, p.v, p.vnx, p.vny, p.nch, p.hwin, p.ncc_v);
        const int win = 2 * p.hwin + 1;
        const size_t lds = sizeof(float) * (size_t)p.nch * win * ((kNccPxb + 2 * p.hwin) + (kNccPxb + p.L - 1 + 2 * p.hwin));
        const dim3 grid((unsigned)(((p.nx + kNccPxb - 1) / kNccPxb) * (long long)p.ny));
        hipError_t e = hipSuccess;
        switch (p.hwin) {
            case 1:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<1>, grid, dim3(256), lds, s, p);
                break;
            case 2:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<2>, grid, dim3(256), lds, s, p);
                break;
            default:
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cost_ncc<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) hipLaunchKernelGGL(k_cost_ncc<3>, grid, dim3(256), lds, s, p);
                break;
        }
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (p.costfn >= 4 && p.ncc_u && p.ncc_v && p.C && !p.C8 && !p.rlo && p.L % 4 == 0) {
        const long long vpix = (long long)p.vnx * p.vny;
        hipLaunchKernelGGL(k_bt_spans, dim3((unsigned)((npix * p.nch + 255) / 256)), dim3(256), 0, s, p.u, p.nx, p.ny, p.nch, p.ncc_u);
        hipLaunchKernelGGL(k_bt_spans, dim3((unsigned)((vpix * p.nch + 255) / 256)), dim3(256), 0, s, p.v, p.vnx, p.vny, p.nch, p.ncc_v);
        long long nw = ((long long)((p.nx + 3) / 4) * p.ny + 3) / 4;
        if (nw > 256 * 64) nw = 256 * 64;
        if (nw < 1) nw = 1;
        if (p.costfn == 5) launch_btx<5>(p, nw, s);
        else launch_btx<4>(p, nw, s);
        return hipGetLastError();
    }
    // differences / multi-word census without a compact form: fp32 volume only (see mgm_costvolume_build_dev)
    if (p.costfn <= 2 && p.C && !p.C8 && !p.rlo && p.L % 4 == 0) {
        long long nw = ((long long)((p.nx + 3) / 4) * p.ny + 3) / 4;
        if (nw > 256 * 64) nw = 256 * 64;
        if (nw < 1) nw = 1;
        if (p.costfn == 0) launch_btx<0>(p, nw, s);
        else if (p.costfn == 1) launch_btx<1>(p, nw, s);
        else launch_btx<2>(p, nw, s);
        return hipGetLastError();
    }
    // (k_cost_diffx takes truncDist = +INF or a non-negative number, sign bit clear; anything else goes to k_cost below)
    if (!p.C && p.C8 && (p.costfn == 0 || p.costfn == 1) && !p.rlo && npix < 0x7fffffffll && c8_supported(p.L) &&
        (p.cbytes == 1 || p.cbytes == 2) && p.L * p.cbytes <= 1024 && p.trunc >= 0.0f && !__builtin_signbit(p.trunc)) {
        if (p.cbytes == 2) p.costfn == 1 ? launch_diffx<2, true>(p, s) : launch_diffx<2, false>(p, s);
        else p.costfn == 1 ? launch_diffx<1, true>(p, s) : launch_diffx<1, false>(p, s);
        return hipGetLastError();
    }
    if (!p.C && p.C8 && p.costfn == 2 && p.nch == 1 && c8_supported(p.L)) {
        const unsigned tb = p.trunc == __builtin_huge_valf() ? 255u : (unsigned)p.trunc;
        long long nb = (npix + 3) / 4;
        if (nb > 256 * 32) nb = 256 * 32;
        const dim3 block(256);
        if (npix < 0x7fffffffll && p.nx % 4 == 0) {  // (every compact label count is a multiple of 16)
            long long nw = (npix * p.L / 4096 + 3) / 4 + 1;
            if (nw > 256 * 32) nw = 256 * 32;
            const dim3 gridw((unsigned)nw);
            switch (p.L) {
                case 64: hipLaunchKernelGGL(k_cost_census8x<64>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 128: hipLaunchKernelGGL(k_cost_census8x<128>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 192: hipLaunchKernelGGL(k_cost_census8x<192>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 256: hipLaunchKernelGGL(k_cost_census8x<256>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 384: hipLaunchKernelGGL(k_cost_census8x<384>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 512: hipLaunchKernelGGL(k_cost_census8x<512>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                case 768: hipLaunchKernelGGL(k_cost_census8x<768>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
                default: hipLaunchKernelGGL(k_cost_census8x<1024>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            }
            return hipGetLastError();
        }
        if ((p.L == 64 || p.L == 128 || p.L == 256 || p.L == 512) && npix < 0x7fffffffll && p.Lreal == p.L) {
            long long nw = (npix * p.L / 1024 + 3) / 4 + 1;
            if (nw > 256 * 32) nw = 256 * 32;
            const dim3 gridw((unsigned)nw);
            switch (p.L) {
                case 64: hipLaunchKernelGGL(k_cost_census8w<64>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                case 128: hipLaunchKernelGGL(k_cost_census8w<128>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                case 256: hipLaunchKernelGGL(k_cost_census8w<256>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
                default: hipLaunchKernelGGL(k_cost_census8w<512>, gridw, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8); break;
            }
            return hipGetLastError();
        }
        const dim3 grid((unsigned)nb);
        switch (p.L / 64) {
            case 1: hipLaunchKernelGGL(k_cost_census8<1>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 2: hipLaunchKernelGGL(k_cost_census8<2>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 3: hipLaunchKernelGGL(k_cost_census8<3>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 4: hipLaunchKernelGGL(k_cost_census8<4>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 6: hipLaunchKernelGGL(k_cost_census8<6>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 8: hipLaunchKernelGGL(k_cost_census8<8>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            case 12: hipLaunchKernelGGL(k_cost_census8<12>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
            default: hipLaunchKernelGGL(k_cost_census8<16>, grid, block, 0, s, p.cu, p.cv, p.nx, p.ny, p.vnx, p.vny, p.dmin, tb, p.C8, p.Lreal); break;
        }
        return hipGetLastError();
    }
    if (p.Lreal != p.L) return hipErrorInvalidValue;  // (padded layouts: only the kernels above write them)
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
