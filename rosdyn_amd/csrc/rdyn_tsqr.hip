// rdyn_tsqr.hip -- tall-skinny QR of the stacked regressor [A | tau_meas] WITHOUT forming A'A (BASELINE.json configs[2]:
// "regressor + TSQR Gram").  The Gram route (rdyn_duo_gram.hip -> pivoted Cholesky) squares the condition number: fine for
// excitation trajectories, useless beyond cond(A) ~ 1e8.  Here the R factor itself is accumulated:
//
//   leaf     every wave keeps a running upper-triangular R (lane j holds column j in registers) and folds row blocks into it by
//            Householder reflections: R <- qr([R ; block]).  A block is the LDS tile of 16 samples that the wave's own row-pair
//            sweep has just produced (rdyn_regressor_tsqr; packed column-major tile of rdyn_lds_gram.hip: the columns of link f
//            only hold the rows of the joints that can be non-zero there, so the reflections skip the structural zeros), or 32
//            rows of a device matrix (rdyn_tsqr).  Per column k: one pass over the block's rows gives |y_k|^2 and y_k . y_j for
//            every lane's column j, the reflection is applied in a second pass; y_k is read by all lanes from one LDS address
//            (broadcast).  No cross-lane reduction, no barrier: a wave never touches another wave's data.
//   tree     k_tsqr_combine folds four R factors into one per wave, level by level in a fixed order (bitwise reproducible),
//            with the same update (an R factor is a block whose column k has k + 1 rows).
//   ranks    the (P + 1)^2 factor of every rank is all-gathered and folded on the host (rdyn_tsqr_combine_host): same payload
//            as the Gram all-reduce.
// Result: R1 = [R d; 0 rho] with A = Q R, d = Q' tau_meas, rho = |residual| -- solve with rdyn_solve_r_factor.
// Cost: ~4 (n = 6) to ~8 (n = 7: 71 columns on 64 lanes, two column slots) times the Gram kernel; it is the robust path, the
// Gram stays the default.  fp64 VALU + LDS only (Householder updates are rank-1: nothing for the matrix cores at this width).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_duo_common.h"

#define DUO_BARRIER()  // the link body shared with rdyn_duo_gram.hip synchronises with its consumer wave there; not here

namespace
{
typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));

__device__ __forceinline__ double lane_value(double x, int src_lane)  // src_lane: compile-time constant after unrolling
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// R <- qr([R ; block]).  NC columns; lane j owns column j (Rc) and, if TWO, column 64 + j (Rc2).  colinfo(k, base, rows): LDS byte
// offset of column k of the block and the (even) number of rows it stores; my_base / my_base2: the same for this lane's columns.
// Columns to the right of k store at least as many rows as column k (prefix property of every block kind used here).
// RQ: the rows of every column are a multiple of RQ (16 for the sweep tiles and the 32-row blocks, 2 for triangular factors): the
// row loops move RQ rows per iteration with all their LDS reads issued up front -- a lone wave is bound by the LDS round trip,
// not by the arithmetic (two rows per trip measured 14 ms at N = 1e6, n = 6).
template <int NC, bool TWO, int RQ, class ColInfo>
__device__ __forceinline__ void tsqr_update(double (&Rc)[NC], double (&Rc2)[TWO ? NC : 1], const char* blk, ColInfo colinfo, int my_base, int my_base2,
                                            int lane)
{
  constexpr int H = RQ / 2;  // 16-byte pairs per iteration
#pragma unroll
  for (int k = 0; k < NC; ++k)
  {
    int base_k, rows;
    colinfo(k, base_k, rows);
    if (rows <= 0) continue;  // wave-uniform
    const char* const yk_p = blk + base_k;
    double sigma = 0.0, d = 0.0, d2 = 0.0;
    for (int r = 0; r < rows; r += RQ)
    {
      d2a yk[H], yj[H], yq[TWO ? H : 1];
#pragma unroll
      for (int h = 0; h < H; ++h)
      {
        yk[h] = *(const d2a*)(yk_p + (r + 2 * h) * 8);
        yj[h] = *(const d2a*)(blk + my_base + (r + 2 * h) * 8);
        if (TWO) yq[h] = *(const d2a*)(blk + my_base2 + (r + 2 * h) * 8);
      }
#pragma unroll
      for (int h = 0; h < H; ++h)
      {
        sigma = fma(yk[h].x, yk[h].x, fma(yk[h].y, yk[h].y, sigma));
        d = fma(yk[h].x, yj[h].x, fma(yk[h].y, yj[h].y, d));
        if (TWO) d2 = fma(yk[h].x, yq[h].x, fma(yk[h].y, yq[h].y, d2));
      }
    }
    if (sigma == 0.0) continue;  // every lane computed the same sigma: uniform
    const double alpha = k < 64 ? lane_value(Rc[k < NC ? k : 0], k & 63) : lane_value(Rc2[TWO ? k : 0], k & 63);
    const double norm = sqrt(fma(alpha, alpha, sigma));
    const double beta = alpha > 0.0 ? -norm : norm;
    const double v0 = alpha - beta;
    const double scale = 2.0 / fma(v0, v0, sigma);
    const bool on = lane > k && lane < NC;
    const bool on2 = TWO && 64 + lane > k && 64 + lane < NC;
    const double f = on ? scale * fma(v0, Rc[k], d) : 0.0;
    const double f2 = on2 ? scale * fma(v0, Rc2[TWO ? k : 0], d2) : 0.0;
    Rc[k] = fma(-f, v0, Rc[k]);
    if (lane == k) Rc[k] = beta;
    if (TWO)
    {
      Rc2[TWO ? k : 0] = fma(-f2, v0, Rc2[TWO ? k : 0]);
      if (64 + lane == k) Rc2[TWO ? k : 0] = beta;
    }
    if (on || on2)
      for (int r = 0; r < rows; r += RQ)
      {
        d2a yk[H], yj[H], yq[TWO ? H : 1];
#pragma unroll
        for (int h = 0; h < H; ++h)
        {
          yk[h] = *(const d2a*)(yk_p + (r + 2 * h) * 8);
          yj[h] = *(const d2a*)(blk + my_base + (r + 2 * h) * 8);
          if (TWO) yq[h] = *(const d2a*)(blk + my_base2 + (r + 2 * h) * 8);
        }
#pragma unroll
        for (int h = 0; h < H; ++h)
        {
          if (on)
          {
            yj[h].x = fma(-f, yk[h].x, yj[h].x);
            yj[h].y = fma(-f, yk[h].y, yj[h].y);
            *(d2a*)(blk + my_base + (r + 2 * h) * 8) = yj[h];
          }
          if (TWO && on2)
          {
            yq[h].x = fma(-f2, yk[h].x, yq[h].x);
            yq[h].y = fma(-f2, yk[h].y, yq[h].y);
            *(d2a*)(blk + my_base2 + (r + 2 * h) * 8) = yq[h];
          }
        }
      }
    wave_lds_fence();  // the next column reads what this step wrote
  }
}

// lane's column(s) of an NC x NC upper-triangular factor, column-major with leading dimension ld
template <int NC, bool TWO>
__device__ __forceinline__ void store_factor(const double (&Rc)[NC], const double (&Rc2)[TWO ? NC : 1], double* out, int ld, int n_cols, int lane)
{
#pragma unroll
  for (int i = 0; i < NC; ++i)
  {
    if (lane < n_cols && i < n_cols) out[(int64_t)lane * ld + i] = i <= lane ? Rc[i] : 0.0;
    if (TWO && 64 + lane < n_cols && i < n_cols) out[(int64_t)(64 + lane) * ld + i] = i <= 64 + lane ? Rc2[TWO ? i : 0] : 0.0;
  }
}

// ---------------------------------------------------------------- leaf: regressor rows from the wave's own sweep
template <int NJ>
__global__ __launch_bounds__(256) void k_regressor_tsqr(const RdynLdsGramArgs fa, double* __restrict__ factors)
{
  constexpr int P = 10 * NJ, NC = P + 1;
  constexpr bool TWO = NC > 64;
  constexpr bool DIRECT = false;
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  ChainPtr c = as_const(fa.chain);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const tile = lds_raw + (size_t)wave * fa.tile_bytes;
  const int n = fa.n_active;
  const int s_loc = lane >> 2, k = lane & 3;
  const int r0 = k, r1 = k + 4;
  int fB = NJ;
  for (int f = NJ - 1; f >= 0; --f)
    if (fa.lds_m[f] >= 5) fB = f;
  // my column(s) of the tile
  auto col_of = [&](int p, int& base, int& rows) {
    if (p < P)
    {
      const int f = p / 10;
      base = fa.lds_off[f] + (p - 10 * f) * fa.lds_stride[f];
      rows = 16 * fa.lds_m[f];
    }
    else
    {
      base = fa.lds_off_b;
      rows = 16 * n;
    }
  };
  int my_base = 0, my_rows = 0, my_base2 = 0, my_rows2 = 0;
  col_of(lane < NC ? lane : NC - 1, my_base, my_rows);
  if (TWO) col_of(64 + lane < NC ? 64 + lane : NC - 1, my_base2, my_rows2);
  (void)my_rows;
  (void)my_rows2;
  double Rc[NC], Rc2[TWO ? NC : 1];
#pragma unroll
  for (int i = 0; i < NC; ++i) Rc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < (TWO ? NC : 1); ++i) Rc2[i] = 0.0;

  const int64_t n_tiles = (fa.n_samples + 15) / 16;
  const int64_t t_first = (int64_t)blockIdx.x * 4 + wave, t_step = (int64_t)gridDim.x * 4;
  for (int64_t tl = t_first; tl < n_tiles; tl += t_step)
  {
    // ---------------- sweep (row-pair lanes, as the sweeper of rdyn_duo_gram.hip): my sample's rows k and k + 4 -> LDS tile
    int64_t sx = tl * 16 + s_loc;
    const bool valid = sx < fa.n_samples;
    if (!valid) sx = fa.n_samples - 1;
    const int64_t o = sx * fa.in_ss;
    double qa = 0.0, dqa = 0.0, ddqa = 0.0, qb = 0.0, dqb = 0.0, ddqb = 0.0, tb0 = 0.0, tb1 = 0.0;
    if (fa.bcol)
    {
      if (r0 < n) tb0 = fa.bcol[o + r0 * fa.in_sj];
      if (r1 < n) tb1 = fa.bcol[o + r1 * fa.in_sj];
    }
    if (k < n)
    {
      qa = fa.q[o + k * fa.in_sj];
      dqa = fa.dq[o + k * fa.in_sj];
      ddqa = fa.ddq[o + k * fa.in_sj];
    }
    if (k + 4 < n)
    {
      qb = fa.q[o + (k + 4) * fa.in_sj];
      dqb = fa.dq[o + (k + 4) * fa.in_sj];
      ddqb = fa.ddq[o + (k + 4) * fa.in_sj];
    }
    if (!valid) tb0 = tb1 = 0.0;
    const int m0idx = valid ? r0 : -2, m1idx = valid ? r1 : -2;
    double sna, csa, snb, csb;
    sincos(qa, &sna, &csa);
    sincos(qb, &snb, &csb);
    const double oca = 1.0 - csa, ocb = 1.0 - csb;
    V3 w = mk(0, 0, 0), vl = mk(0, 0, 0), al = mk(0, 0, 0);
    V3 lin = mk(-c->g[0], -c->g[1], -c->g[2]);
    V3 L0 = mk(0, 0, 0), A0 = mk(0, 0, 0), L1 = mk(0, 0, 0), A1 = mk(0, 0, 0);
#pragma unroll 1
    for (int f = 0; f < NJ; ++f)
    {
#include "rdyn_duo_link_body.inc"
    }
    {
      char* const lb = tile + fa.lds_off_b + s_loc * 8;
      if (r0 < n) *(double*)(lb + r0 * 128) = tb0;
      if (r1 < n) *(double*)(lb + r1 * 128) = tb1;
    }
    wave_lds_fence();
    // ---------------- fold the tile into the running factor
    tsqr_update<NC, TWO, 16>(Rc, Rc2, tile, col_of, my_base, my_base2, lane);
  }
  store_factor<NC, TWO>(Rc, Rc2, factors + ((int64_t)blockIdx.x * 4 + wave) * (NC * NC), NC, NC, lane);
}

// ---------------------------------------------------------------- leaf: 32-row blocks of a column-major device matrix
template <int NC>
__global__ __launch_bounds__(256) void k_tsqr_rows(const double* __restrict__ A, const double* __restrict__ b, int64_t rows, int64_t lda, int n_cols,
                                                   double* __restrict__ factors)
{
  constexpr int TR = 32, SB = (TR + 2) * 8;  // block: NC columns of TR rows (+ pad), column-major
  extern __shared__ __attribute__((aligned(32))) char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const blk = lds_raw + (size_t)wave * (NC * SB);
  double Rc[NC], Rc2[1] = {0.0};
#pragma unroll
  for (int i = 0; i < NC; ++i) Rc[i] = 0.0;
  const int nc1 = n_cols + (b ? 1 : 0);  // the right-hand side rides as one more column
  auto col_of = [&](int kcol, int& base, int& nrows) {
    base = kcol * SB;
    nrows = kcol < nc1 ? TR : 0;
  };
  const int my_base = (lane < NC ? lane : NC - 1) * SB;
  for (int i = lane * 8; i < NC * SB; i += 64 * 8) *(double*)(blk + i) = 0.0;  // columns >= nc1 are never loaded: keep them zero
  wave_lds_fence();
  const int64_t n_blocks = (rows + TR - 1) / TR;
  const int64_t b_first = (int64_t)blockIdx.x * 4 + wave, b_step = (int64_t)gridDim.x * 4;
  const int half = lane >> 5, rl = lane & 31;
  for (int64_t bi = b_first; bi < n_blocks; bi += b_step)
  {
    const int64_t r = bi * TR + rl;
    for (int cc = 0; cc < nc1; cc += 2)  // two columns per instruction: lanes 0-31 / 32-63
    {
      const int col = cc + half;
      double v = 0.0;
      if (col < nc1 && r < rows) v = col < n_cols ? A[(int64_t)col * lda + r] : b[r];
      if (col < NC) *(double*)(blk + col * SB + rl * 8) = v;
    }
    wave_lds_fence();
    tsqr_update<NC, false, 16>(Rc, Rc2, blk, col_of, my_base, 0, lane);
  }
  store_factor<NC, false>(Rc, Rc2, factors + ((int64_t)blockIdx.x * 4 + wave) * (NC * NC), NC, NC, lane);
}

// ---------------------------------------------------------------- tree: every wave folds up to four factors into one
template <int NC>
__global__ __launch_bounds__(64) void k_tsqr_combine(const double* __restrict__ in, int count, double* __restrict__ out, int out_ld, int out_cols,
                                                     const double* __restrict__ extra /* one more factor (accumulate), or null */, int extra_ld)
{
  constexpr bool TWO = NC > 64;
  constexpr int SB = ((NC + 3) & ~1) * 8;
  extern __shared__ __attribute__((aligned(32))) char blk[];
  const int lane = threadIdx.x;
  const int first = blockIdx.x * 4;
  double Rc[NC], Rc2[TWO ? NC : 1];
  // the first factor becomes the running one
  {
    const double* f0 = in + (int64_t)first * (NC * NC);
#pragma unroll
    for (int i = 0; i < NC; ++i)
    {
      Rc[i] = (lane < NC && i <= lane) ? f0[(int64_t)lane * NC + i] : 0.0;
      if (TWO) Rc2[TWO ? i : 0] = (64 + lane < NC && i <= 64 + lane) ? f0[(int64_t)(64 + lane) * NC + i] : 0.0;
    }
    if (!TWO) Rc2[0] = 0.0;
  }
  auto col_of = [&](int kcol, int& base, int& nrows) {
    base = kcol * SB;
    nrows = (kcol + 2) & ~1;  // column k of an upper-triangular factor: rows 0 .. k (+ one zero row to make it even)
  };
  const int my_base = (lane < NC ? lane : NC - 1) * SB, my_base2 = (64 + lane < NC ? 64 + lane : NC - 1) * SB;
  const int n_more = (count - first < 4 ? count - first : 4) - 1;
  for (int t = 1; t <= n_more + (extra && blockIdx.x == 0 ? 1 : 0); ++t)
  {
    const bool is_extra = t > n_more;
    const double* ft = is_extra ? extra : in + (int64_t)(first + t) * (NC * NC);
    const int ld = is_extra ? extra_ld : NC;
    const int cols = is_extra ? out_cols : NC;
    for (int col = 0; col < NC; ++col)
      for (int i = lane; i < NC + 1; i += 64)
        *(double*)(blk + col * SB + i * 8) = (col < cols && i <= col) ? ft[(int64_t)col * ld + i] : 0.0;
    wave_lds_fence();
    tsqr_update<NC, TWO, 2>(Rc, Rc2, blk, col_of, my_base, my_base2, lane);
  }
  store_factor<NC, TWO>(Rc, Rc2, out + (int64_t)blockIdx.x * (NC * NC), out_ld, out_cols, lane);  // the last level is one wave: offset 0
}

template <class K>
hipError_t opt_in_lds(K kernel, std::atomic<uint64_t>& done)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit))
  {
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  return hipSuccess;
}

// folds `count` NC x NC factors at `slab` down to one, written to R (ld_out x n_out, column-major); scratch = second slab region
template <int NC>
hipError_t combine_tree(double* slab, int count, double* scratch, double* R, int n_out, const double* extra, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds(k_tsqr_combine<NC>, attr);
  if (e != hipSuccess) return e;
  const size_t lds = (size_t)NC * (((NC + 3) & ~1) * 8);
  double* in = slab;
  double* out = scratch;
  while (count > 4)
  {
    const int nout = (count + 3) / 4;
    hipLaunchKernelGGL((k_tsqr_combine<NC>), dim3(nout), dim3(64), lds, st, in, count, out, NC, NC, nullptr, 0);
    count = nout;
    double* t = in;
    in = out;
    out = t;
  }
  // last level: straight into the caller's buffer (compact n_out x n_out), folding the caller's previous factor if accumulating
  hipLaunchKernelGGL((k_tsqr_combine<NC>), dim3(1), dim3(64), lds, st, in, count, R, n_out, n_out, extra, n_out);
  return hipGetLastError();
}

template <int NJ>
hipError_t launch_regressor_tsqr(const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, double* slab, double* scratch, double* R, const double* extra,
                                 hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds(k_regressor_tsqr<NJ>, attr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_regressor_tsqr<NJ>), dim3(blocks), dim3(256), lds_bytes, st, a, slab);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return combine_tree<10 * NJ + 1>(slab, blocks * 4, scratch, R, 10 * NJ + 1, extra, st);
}

template <int NC>
hipError_t launch_tsqr_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* slab, double* scratch, double* R,
                            const double* extra, hipStream_t st)
{
  static std::atomic<uint64_t> attr{0};
  hipError_t e = opt_in_lds(k_tsqr_rows<NC>, attr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tsqr_rows<NC>), dim3(blocks), dim3(256), (size_t)4 * NC * (34 * 8), st, A, b, rows, lda, n_cols, slab);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return combine_tree<NC>(slab, blocks * 4, scratch, R, n_cols + (b ? 1 : 0), extra, st);
}
}  // namespace

// factors per launch and doubles of workspace (two slab regions: leaves + tree levels)
int rdyn_tsqr_padded_cols(int n_cols_with_rhs) { return n_cols_with_rhs <= 16 ? 16 : n_cols_with_rhs <= 32 ? 32 : n_cols_with_rhs <= 48 ? 48 : n_cols_with_rhs <= 64 ? 64 : 0; }
size_t rdyn_tsqr_workspace_doubles(int nc, int blocks) { return (size_t)(blocks * 4 + blocks + 4) * nc * nc; }

hipError_t rdyn_launch_regressor_tsqr(int n_joints, const RdynLdsGramArgs& a, int blocks, size_t lds_bytes, double* workspace, double* R, int accumulate,
                                      hipStream_t st)
{
  const int nc = 10 * n_joints + 1;
  double* slab = workspace;
  double* scratch = workspace + (size_t)blocks * 4 * nc * nc;
  const double* extra = accumulate ? R : nullptr;
  switch (n_joints)
  {
  case 2: return launch_regressor_tsqr<2>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  case 3: return launch_regressor_tsqr<3>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  case 4: return launch_regressor_tsqr<4>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  case 5: return launch_regressor_tsqr<5>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  case 6: return launch_regressor_tsqr<6>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  case 7: return launch_regressor_tsqr<7>(a, blocks, lds_bytes, slab, scratch, R, extra, st);
  default: return hipErrorInvalidValue;
  }
}

hipError_t rdyn_launch_tsqr_rows(const double* A, const double* b, int64_t rows, int64_t lda, int n_cols, int blocks, double* workspace, double* R,
                                 int accumulate, hipStream_t st)
{
  const int nc = rdyn_tsqr_padded_cols(n_cols + (b ? 1 : 0));
  double* slab = workspace;
  double* scratch = workspace + (size_t)blocks * 4 * nc * nc;
  const double* extra = accumulate ? R : nullptr;
  switch (nc)
  {
  case 16: return launch_tsqr_rows<16>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st);
  case 32: return launch_tsqr_rows<32>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st);
  case 48: return launch_tsqr_rows<48>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st);
  case 64: return launch_tsqr_rows<64>(A, b, rows, lda, n_cols, blocks, slab, scratch, R, extra, st);
  default: return hipErrorInvalidValue;
  }
}
