// rdyn_gram_common.h -- pieces shared by the Gram kernels (rdyn_gram.hip, rdyn_fused_gram.hip, rdyn_lds_gram.hip,
// rdyn_pipe_gram.hip)
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

// Epilogue of every Gram kernel (256-thread workgroups = 4 waves): the per-wave accumulator tiles are summed through LDS
// in wave order (fixed order: bitwise reproducible) and the workgroup's NT tiles go to its own slab -- overwritten, or
// added to what the slab holds when `accumulate` (chunked two-kernel path).  `red` = NT * 256 doubles of LDS that are no
// longer in use; lane (cl = lane & 15, g = lane >> 4) of `wave`.  C/D layout of v_mfma_f64_16x16x4_f64:
// row = (lane >> 4) + 4 * reg, col = lane & 15.  Ends with the slab written (no barrier after it).
template <int NT>
__device__ __forceinline__ void gram_block_reduce_to_slab(const d4* acc, double* red, int wave, int cl, int g, double* slab, bool accumulate)
{
  for (int w = 0; w < 4; ++w)
  {
    if (wave == w)
    {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          const int idx = t * 256 + ((g + 4 * r) * 16 + cl);
          red[idx] = (w == 0) ? acc[t][r] : red[idx] + acc[t][r];
        }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < NT * 256; i += 256) slab[i] = accumulate ? slab[i] + red[i] : red[i];
}
}  // namespace
#endif
