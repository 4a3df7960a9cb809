// rdyn_gram.hip -- normal equations of the stacked regressor on the fp64 matrix cores (gfx950).
//
// No reference counterpart: rosdyn_core only produces getRegressor rows (primitives_impl.h:1295-1355); the
// least-squares step that consumed them lived in the external rosdyn_identification (README.md:15).
// BASELINE.json's north star asks for it: "MFMA used only for the final tall-skinny (N*ndof x 10*nlinks)
// regressor Gram matrix".
//
//   Given A (rows x P, column-major, leading dimension lda) and b (rows), accumulate
//       G = [A b]^T [A b]   ->  A^T A (P x P),  A^T b (P),  b^T b
//   with v_mfma_f64_16x16x4_f64:  D(16x16) += Aop(16x4) * Bop(4x16),  Aop[i][k] = A[row k][16 rb + i],
//   Bop[k][j] = A[row k][16 cb + j].  Only the NB(NB+1)/2 upper tiles are computed.
//
// Operand feed without any transposition: lane l (c = l & 15, g = l >> 4) loads FOUR CONSECUTIVE ROWS
// (32 contiguous bytes) of column 16 cb + c starting at row r0 + 4 g; k-step t (t = 0..3) then uses element t of
// every lane, i.e. rows {r0 + 4 g + t}: a permutation of which rows share a k-step, which a sum over rows does
// not care about.  Per column the four lane groups read one full 128-byte line.
//
// Structure: in the element-major regressor image the rows of input joint j are exactly zero left of column
// 10 * (chain index of j) (block upper-triangular Y, primitives_impl.h:1341-1347).  Column blocks that lie
// entirely in that zero band are neither loaded nor multiplied: for n = 6 / P = 60 that removes 40 % of the MFMAs
// and of the HBM reads, for n = 7 / P = 70 41 %.
//
// Work split: each wave walks 16-row groups with stride (#waves * 16); per-wave tiles are summed over the block
// in LDS, every block adds into ITS OWN slab of the workspace (no atomics, bitwise reproducible), and
// k_gram_finish sums the slabs in fixed order and writes the symmetric result.
#include <hip/hip_runtime.h>
#include "rdyn_kernels.h"
#include "rdyn_gram_common.h"

namespace
{

// BANDS: the rows come in blocks with a leading zero band (a.row_block > 0: the chunk images of a regressor) -- one straight-line MFMA
// block per possible band behind a wave-uniform switch.  A plain matrix (rdyn_gram, the subsample of rdyn_tsqr) takes the instantiation
// without the switch: where its branches meet the compiler copies accumulators (504 B of scratch and 152 us for the subsample pass at
// seven column blocks).
template <int NB, bool BANDS>
__global__ __launch_bounds__(256) void k_gram(const RdynGramArgs a)
{
  constexpr int NT = NB * (NB + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: row bookkeeping stays on the SALU
  const int c = lane & 15, g = lane >> 4;
  const int64_t R = a.rows;

  // per-lane column base pointers (null -> column of zeros: padding beyond P + 1)
  const double* col[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb)
  {
    const int p = 16 * cb + c;
    col[cb] = (p < a.P) ? a.A + (int64_t)p * a.lda : ((p == a.P && a.b) ? a.b : nullptr);
  }

  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

  const int64_t gmul = a.group_stride > 1 ? a.group_stride : 1;  // subsample: every gmul-th 16-row group
  const int64_t wstride = (int64_t)gridDim.x * 4 * 16 * gmul;
  int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16 * gmul;

  // first column block that can be non-zero for the 16-row group starting at r (wave-uniform)
  // (rows only grow, so the row-block index is tracked incrementally: no 64-bit division per group)
  int jb = 0;
  int64_t bound = a.row_block;  // rows < bound belong to row block jb
  auto cb_min_of = [&](int64_t r) -> int {
    if (!BANDS || a.row_block <= 0) return 0;
    while (r >= bound)
    {
      ++jb;
      bound += a.row_block;
    }
    // the group may straddle several row blocks (row_block < 16: short last chunk) and first_col is not monotonic
    // when the input joints are not in chain order: take the minimum over every row block the 16 rows touch
    int fc = a.first_col[jb];
    int j2 = jb + 1;
    for (int64_t b2 = bound; b2 <= r + 15 && b2 < R && j2 < RDYN_MAX_SWEPT_JOINTS; b2 += a.row_block, ++j2)
      if (a.first_col[j2] < fc) fc = a.first_col[j2];
    return fc >> 4;
  };

  auto load = [&](int64_t rbase, int cbm, d4* v) {
    const int64_t r = rbase + 4 * g;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
    {
      d4 x = (d4){0.0, 0.0, 0.0, 0.0};
      if (cb >= cbm && col[cb])
      {
        if (r + 4 <= R)
          x = *(const d4u*)(col[cb] + r);
        else
        {
          if (r + 0 < R) x[0] = col[cb][r + 0];
          if (r + 1 < R) x[1] = col[cb][r + 1];
          if (r + 2 < R) x[2] = col[cb][r + 2];
        }
      }
      v[cb] = x;
    }
  };

  d4 cur[NB], nxt[NB];
  int cbm = 0, cbm_n = 0;
  if (r0 < R)
  {
    cbm = cb_min_of(r0);
    load(r0, cbm, cur);
  }
  while (r0 < R)
  {
    const int64_t rn = r0 + wstride;
    if (rn < R)
    {
      cbm_n = cb_min_of(rn);
      load(rn, cbm_n, nxt);  // prefetch the next 16-row group behind this group's MFMAs
    }
    // one straight-line MFMA block per possible zero band (wave-uniform switch, no per-tile branches)
    switch ((BANDS && NB > 1) ? cbm : 0)
    {
    case 0: mfma_group<NB, 0>(cur, acc); break;
    case 1: mfma_group<NB, 1>(cur, acc); break;
    case 2: mfma_group<NB, 2>(cur, acc); break;
    case 3: mfma_group<NB, 3>(cur, acc); break;
    case 4: mfma_group<NB, 4>(cur, acc); break;
    case 5: mfma_group<NB, 5>(cur, acc); break;
    default: mfma_group<NB, 6>(cur, acc); break;
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) cur[cb] = nxt[cb];
    cbm = cbm_n;
    r0 = rn;
  }

  // ---- block reduction in LDS, then add into this block's slab
  __shared__ double red[NT * 256];
  gram_block_reduce_to_slab<NT>(acc, red, wave, c, g, a.slabs + (int64_t)blockIdx.x * (NT * 256), a.accumulate != 0);
}

// sums the per-block slabs (fixed order) and scatters the tiles into G (P x P, both triangles), c = A^T b, bb.
// One workgroup per 32 tile elements: FG slab groups x 32 elements, slab groups reduced through LDS in fixed order
// (FG = 32: 1 024 threads, 8 slabs each at 256 slabs -- 7.8 MB of slabs at 6 joints in ~5 us; 8 groups: 13 us).
#ifndef RDYN_GRAM_FINISH_GROUPS
#define RDYN_GRAM_FINISH_GROUPS 32
#endif
constexpr int FG = RDYN_GRAM_FINISH_GROUPS;
__global__ __launch_bounds__(32 * FG) void k_gram_finish(const RdynGramArgs a, int nb, int n_slabs)
{
  if (a.run_flag && *a.run_flag == 0) return;
  const int nt = nb * (nb + 1) / 2;
  const int e_loc = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e_loc;
  __shared__ double part[FG][32];
  double s = 0.0;
  if (i < nt * 256)
    for (int b = grp; b < n_slabs; b += FG) s += a.slabs[(int64_t)b * nt * 256 + i];
  part[grp][e_loc] = s;
  __syncthreads();
  if (grp != 0 || i >= nt * 256) return;
  s = 0.0;
  for (int k = 0; k < FG; ++k) s += part[k][e_loc];
  const int t = i >> 8, e = i & 255;
  // tile t -> (rb, cb): t = cb (cb + 1) / 2 + rb
  int cb = 0;
  while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
  const int rb = t - cb * (cb + 1) / 2;
  const int pp1 = 16 * rb + (e >> 4), pp2 = 16 * cb + (e & 15);  // positions in the slabs' column order
  const int P = a.P;
  // slab column -> regressor column (P = the measured-torque column, > P = padding)
  auto col_of = [&](int pp) -> int {
    if (a.col_shift > 0) return pp < a.col_shift ? P + 1 : pp - a.col_shift;  // padding in front of the natural order
    if (a.desc_nj <= 0) return pp;
    // [desc_k component columns | tau_meas | link desc_nj - 1 | ... | link 0 | padding]; in G the links come first, then the components
    if (pp < a.desc_k) return P - a.desc_k + pp;
    if (pp == a.desc_k) return P;
    const int pl = pp - a.desc_k - 1;
    if (pl >= 10 * a.desc_nj) return P + 1;
    return 10 * (a.desc_nj - 1 - pl / 10) + pl % 10;
  };
  const int p1 = col_of(pp1), p2 = col_of(pp2);
  const double prev_scale = a.add_to_output ? 1.0 : 0.0;
  if (rb == cb && pp1 > pp2) return;  // diagonal tiles hold both triangles: keep the upper one
  if (p1 < P && p2 < P)
  {
    a.G[(int64_t)p2 * P + p1] = prev_scale * a.G[(int64_t)p2 * P + p1] + s;
    if (p1 != p2) a.G[(int64_t)p1 * P + p2] = prev_scale * a.G[(int64_t)p1 * P + p2] + s;
  }
  else if ((p2 == P && p1 < P) || (p1 == P && p2 < P))
  {
    const int pc = p1 < P ? p1 : p2;
    if (a.c) a.c[pc] = prev_scale * a.c[pc] + s;
  }
  else if (p1 == P && p2 == P)
  {
    if (a.bb) a.bb[0] = prev_scale * a.bb[0] + s;
  }
}

// Normal equations of the reduced chain -> normal equations of the chain itself (rdyn_chain.hpp: Y = Y_red E, so G = E' G_red E,
// c = E' c_red, bb unchanged).  E is block sparse: the ten columns of link f are X_f applied to the ten columns of reduced link
// red_of[f] (nothing for links upstream of the first input joint); component columns (K of them, behind the link columns) map
// one to one.  One thread per entry of the output, <= 100 fmas each.
__global__ __launch_bounds__(256) void k_gram_expand(const RdynGramExpandArgs a)
{
  const int P = 10 * a.n_joints, Pr = 10 * a.n_red, C = P + a.n_comp_cols, Cr = Pr + a.n_comp_cols;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const double prev_scale = a.add_to_output ? 1.0 : 0.0;
  if (idx >= C * (C + 1))
  {
    if (idx == C * (C + 1) && a.bb) a.bb[0] = prev_scale * a.bb[0] + a.bb_red[0];
    return;
  }
  const int i = idx % C, j = idx / C;  // j == C: the right-hand side c
  // column i of the output = sum over the rows (r0 .. r0 + nr) of the reduced column space with weights wi
  auto rows_of = [&](int col, int& r0, int& nr, const double*& w, int& wstride) {
    if (col >= P)
    {
      r0 = Pr + (col - P);
      nr = 1;
      w = nullptr;
      wstride = 0;
      return;
    }
    const int f = col / 10, p = col - 10 * f, r = a.red_of[f];
    r0 = 10 * (r < 0 ? 0 : r);
    nr = r < 0 ? 0 : 10;
    w = a.X + f * 100 + p;  // X_f(a, p), a = 0..9, stride 10
    wstride = 10;
  };
  int ri, ni, si;
  const double* wi;
  rows_of(i, ri, ni, wi, si);
  double s = 0.0;
  if (j == C)
  {
    if (!a.c) return;
    for (int x = 0; x < ni; ++x) s = fma(wi ? wi[x * si] : 1.0, a.c_red[ri + x], s);
    a.c[i] = prev_scale * a.c[i] + s;
    return;
  }
  int rj, nj, sj;
  const double* wj;
  rows_of(j, rj, nj, wj, sj);
  for (int y = 0; y < nj; ++y)
  {
    double t = 0.0;
    for (int x = 0; x < ni; ++x) t = fma(wi ? wi[x * si] : 1.0, a.G_red[(int64_t)(rj + y) * Cr + ri + x], t);
    s = fma(wj ? wj[y * sj] : 1.0, t, s);
  }
  a.G[(int64_t)j * C + i] = prev_scale * a.G[(int64_t)j * C + i] + s;
}

__global__ void k_set_double(double* p, double v) { *p = v; }

template <int NB>
hipError_t launch_gram_nb(const RdynGramArgs& a, int blocks, hipStream_t st)
{
  if (a.row_block > 0)
    hipLaunchKernelGGL((k_gram<NB, true>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_gram<NB, false>), dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace

int rdyn_gram_blocks_for(int P) { return (P + 1 + 15) / 16; }

hipError_t rdyn_launch_gram(const RdynGramArgs& a, int blocks, hipStream_t st)
{
  const int nb = rdyn_gram_blocks_for(a.P);
  switch (nb)
  {
  case 1: return launch_gram_nb<1>(a, blocks, st);
  case 2: return launch_gram_nb<2>(a, blocks, st);
  case 3: return launch_gram_nb<3>(a, blocks, st);
  case 4: return launch_gram_nb<4>(a, blocks, st);
  case 5: return launch_gram_nb<5>(a, blocks, st);
  case 6: return launch_gram_nb<6>(a, blocks, st);
  case 7: return launch_gram_nb<7>(a, blocks, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t rdyn_launch_gram_finish(const RdynGramArgs& a, int blocks, hipStream_t st)
{
  const int nb = a.slab_nb > 0 ? a.slab_nb : rdyn_gram_blocks_for(a.P);
  const int nt = nb * (nb + 1) / 2;
  hipLaunchKernelGGL(k_gram_finish, dim3((nt * 256 + 31) / 32), dim3(32 * FG), 0, st, a, nb, blocks);
  return hipGetLastError();
}

hipError_t rdyn_launch_gram_expand(const RdynGramExpandArgs& a, hipStream_t st)
{
  const int C = 10 * a.n_joints + a.n_comp_cols;
  hipLaunchKernelGGL(k_gram_expand, dim3((C * (C + 1) + 1 + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

namespace
{
__global__ __launch_bounds__(256) void k_add_doubles(double* __restrict__ y, const double* __restrict__ x, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] += x[i];
}
}  // namespace
hipError_t rdyn_launch_add_doubles(double* y, const double* x, int64_t n, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_add_doubles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, n);
  return hipGetLastError();
}

hipError_t rdyn_launch_set_double(double* p, double v, hipStream_t st)
{
  hipLaunchKernelGGL(k_set_double, dim3(1), dim3(1), 0, st, p, v);
  return hipGetLastError();
}
