/*
 * ref_post_harness.cc -- TEST INFRASTRUCTURE ONLY (builds into oracle/_ref/libmgm_refpost.so).
 *
 * The steps the reference's main() applies to the disparity maps right after the path -- median_filter
 * (img_tools.h:203-238), leftright_test (mgm.cc:68-91), update_dmin_dmax (mgm.cc:120-158) -- behind dense-array entry
 * points, so that mgm_post.hip can be compared with the reference ITSELF on inputs the command line cannot produce at
 * will (NaN labels, +INF costs).  leftright_test and update_dmin_dmax live in the reference's mgm.cc next to main():
 * that file is #included here where it lies, with its main() renamed; no reference source is copied.
 * Built by oracle/Makefile only when $(REF) exists.
 */
#define main mgm_reference_main_unused
#include "mgm.cc"
#undef main

extern "C" {

/* median_filter (img_tools.h:203-238) */
int refpost_median(const float *u, int nx, int ny, int nch, int radius, float *out)
{
    Img I(const_cast<float *>(u), nx, ny, nch);
    Img M = median_filter(I, radius);
    memcpy(out, &M.data[0], sizeof(float) * (size_t)nx * ny * nch);
    return 0;
}

/* leftright_test (mgm.cc:68-91): dx is modified in place */
int refpost_leftright(float *dx, int nx, int ny, const float *Rdx, int Rnx, int Rny, float threshold)
{
    Img A(dx, nx, ny, 1), B(const_cast<float *>(Rdx), Rnx, Rny, 1);
    leftright_test(A, B, threshold);
    memcpy(dx, &A.data[0], sizeof(float) * (size_t)nx * ny);
    return 0;
}

/* update_dmin_dmax (mgm.cc:120-158) followed by the two remove_nonfinite_values_Img calls of main() (387-388) */
int refpost_update_ranges(const float *outoff, int nx, int ny, float *dminI, float *dmaxI, int slack, int radius)
{
    Img O(const_cast<float *>(outoff), nx, ny, 1), A(dminI, nx, ny, 1), B(dmaxI, nx, ny, 1);
    std::pair<float, float> gm = update_dmin_dmax(O, &A, &B, slack, radius);
    remove_nonfinite_values_Img(A, gm.first);
    remove_nonfinite_values_Img(B, gm.second);
    memcpy(dminI, &A.data[0], sizeof(float) * (size_t)nx * ny);
    memcpy(dmaxI, &B.data[0], sizeof(float) * (size_t)nx * ny);
    return 0;
}

} /* extern "C" */
