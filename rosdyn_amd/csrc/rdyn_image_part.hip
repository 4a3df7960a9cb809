// rdyn_image_part.hip -- one slice of the LDS-staged regressor kernels (rdyn_image_impl.h): every compiled fixed-joint pattern of
// chains with RDYN_IMAGE_NA input joints; RDYN_IMAGE_MULTI = 1 builds the mixed-chain plan kernels (blockIdx.y = item) instead of
// the single-chain ones.  The Makefile compiles this file once per (NA, MULTI) so that the slices build in parallel.
#ifndef RDYN_IMAGE_NA
#error "compile with -DRDYN_IMAGE_NA=<input joints> -DRDYN_IMAGE_MULTI=<0|1>"
#endif
#include "rdyn_image_impl.h"

#define RDYN_CAT2(a, b) a##b
#define RDYN_CAT(a, b) RDYN_CAT2(a, b)

#if RDYN_IMAGE_MULTI
hipError_t RDYN_CAT(rdyn_image_launch_multi_na, RDYN_IMAGE_NA)(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* table, int n_items,
                                                              int64_t max_samples, hipStream_t st)
{
  return image_launch_na<RDYN_IMAGE_NA, true>(n_joints, fix, stacked, table, n_items, max_samples, st);
}
#else
hipError_t RDYN_CAT(rdyn_image_launch_na, RDYN_IMAGE_NA)(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* a, hipStream_t st, int mapped)
{
  return image_launch_na<RDYN_IMAGE_NA, false>(n_joints, fix, stacked, a, 1, a->n_samples, st, mapped);
}
#endif
