// mgm_cost_common.h -- device helpers shared by the K2 kernels of mgm_cost.hip (the general kernel, K1, K7, the format
// conversions) and mgm_cost_fast.hip (the restructured kernels of the costs the hot paths use).
#pragma once
#include "mgm_device.h"

namespace mgm {

__device__ __forceinline__ bool finite_bits(float x)
{
    return (__builtin_bit_cast(unsigned, x) & 0x7f800000u) != 0x7f800000u;
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
// Raise a bit of a flag word that many waves may want to raise: look first -- an atomic per wave on ONE address serialises
// (round 4: RGB absolute differences exceed 254 at nearly every pixel, and 2 M atomicOr on the "no compact form" word made
// K2 take 21.8 ms at 1920x1080x256 where the grey-level volume took 1.55).
__device__ __forceinline__ void flag_once(unsigned *word, unsigned bit)
{
    if ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) == 0u) atomicOr(word, bit);
}

}  // namespace mgm
