// rdyn_duo_common.h -- helpers of the row-pair sweeper body (rdyn_duo_link_body.inc) shared by rdyn_duo_gram.hip and rdyn_tsqr.hip
#ifndef RDYN_DUO_COMMON_H
#define RDYN_DUO_COMMON_H
#include <hip/hip_runtime.h>

namespace
{
// value of lane K of my quad (K a constant after unrolling): two v_mov_b32 with a quad_perm DPP control
template <int K>
__device__ __forceinline__ double quad_bcast_k(double x)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), K * 0x55, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), K * 0x55, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_bcast(double x, int k)
{
  switch (k)
  {
  case 0: return quad_bcast_k<0>(x);
  case 1: return quad_bcast_k<1>(x);
  case 2: return quad_bcast_k<2>(x);
  default: return quad_bcast_k<3>(x);
  }
}
// LDS byte offset of link f's first column when every chain joint is an input joint (rows j <= f stored: stride (16 (f + 1) + pad) * 8)
__device__ __forceinline__ constexpr int duo_direct_off(int f, int pad = 4)
{
  int off = 0;
  for (int g = 0; g < f; ++g) off += 10 * (16 * (g + 1) + pad) * 8;
  return off;
}
// element offsets of the inputs of the rows that lane k of a quad owns (rows k and k + 4): in_map[row] * in_sj (rdyn_kernels.h: in_map)
#define RDYN_DUO_INPUT_OFFSETS(fa, k, oa, ob)                                                                                       \
  const int64_t oa = (int64_t)((k) == 0 ? (fa).in_map[0] : ((k) == 1 ? (fa).in_map[1] : ((k) == 2 ? (fa).in_map[2] : (fa).in_map[3]))) * (fa).in_sj; \
  const int64_t ob = (int64_t)((k) == 0 ? (fa).in_map[4] : ((k) == 1 ? (fa).in_map[5] : ((k) == 2 ? (fa).in_map[6] : (fa).in_map[7]))) * (fa).in_sj
#ifndef RDYN_DUO_TILE_PAD
#define RDYN_DUO_TILE_PAD 4  // (a kernel that sweeps into the compact tile redefines it around its include of rdyn_duo_link_body.inc)
#endif

// ocml's sincos for angles beyond the range of rdyn_sincos_small, out of line: one copy of its Payne-Hanek path in a kernel, not one per
// joint (the one-lane-per-sample sweepers, rdyn_kin_sweepers.inc)
struct DuoSinCos
{
  double sn, cs;
};
__device__ __attribute__((noinline)) inline DuoSinCos duo_sincos_cold(double x)
{
  DuoSinCos r;
  sincos(x, &r.sn, &r.cs);
  return r;
}

}  // namespace
#endif
