// rdyn_gram_common.h -- pieces shared by the Gram kernels (rdyn_gram.hip, rdyn_fused_gram.hip)
#ifndef RDYN_GRAM_COMMON_H
#define RDYN_GRAM_COMMON_H
#include <hip/hip_runtime.h>

namespace
{
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d4u __attribute__((ext_vector_type(4), aligned(8)));

// the 4 k-steps of one 16-row group for all upper tiles (rb <= cb) whose column blocks are >= CBM
template <int NB, int CBM>
__device__ __forceinline__ void mfma_group(const d4* cur, d4* acc)
{
#pragma unroll
  for (int t = 0; t < 4; ++t)
  {
    int ti = 0;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int rb = 0; rb <= cb; ++rb)
      {
        if (rb >= CBM) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[rb][t], cur[cb][t], acc[ti], 0, 0, 0);
        ++ti;
      }
  }
}
}  // namespace
#endif
