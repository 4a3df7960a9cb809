// rdyn_api.cpp -- batched C-ABI entry points (include/rdyn.h): argument checks, layout -> strides,
// chain-constant residency per device, kernel launches.  No host<->device copies of batch data, no
// allocation and no synchronisation after a chain's first use on a device.
//
// Diagnostic environment switches exist ONLY in builds compiled with -DRDYN_ENABLE_PROBES (tools/build_variant.sh; A/B
// measurements, tools/probe_*.py and tools/kbench).  The shipped library never reads the environment: a variable left over
// from a profiling shell cannot change results.
//   RDYN_NO_ROWPAIR=1       regressor in row-contiguous layouts through the one-thread-per-sample kernel
//   RDYN_GRAM_PATH=pipe|lds0|image|two   regressor->Gram kernel: single-wave pipelined LDS tile / two-phase LDS tile /
//                           global image / two kernels (default: wave-pair kernel where eligible, then the single-wave LDS
//                           kernels, then the global image)
//   RDYN_GRAM_UNFUSED=1     same as RDYN_GRAM_PATH=two
//   RDYN_FUSED_BLOCKS=n     persistent workgroups of the fused Gram kernels (default 256 = one per CU)
//   RDYN_FUSED_DEBUG=bits   phase ablation of the fused Gram kernels (timing only: results are then wrong)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include <hip/hip_runtime.h>

#include "rdyn_chain.hpp"
#include "rdyn_kernels.h"

namespace
{

#ifdef RDYN_ENABLE_PROBES
const char* probe_env(const char* name) { return getenv(name); }
#else
const char* probe_env(const char*) { return nullptr; }
#endif

#define RDYN_HIP_TRY(expr)                                                     \
  do                                                                           \
  {                                                                            \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
    {                                                                          \
      rdyn_set_error("HIP error: %s (%s)", hipGetErrorString(_e), #expr);      \
      return RDYN_ERR_HIP;                                                     \
    }                                                                          \
  } while (0)

struct DeviceGuard
{
  int prev = -1;
  bool switched = false;
  int enter(int device)
  {
    if (hipGetDevice(&prev) != hipSuccess)
    {
      rdyn_set_error("no HIP device available (hipGetDevice failed)");
      return RDYN_ERR_NO_DEVICE;
    }
    if (device >= 0 && device != prev)
    {
      if (hipSetDevice(device) != hipSuccess)
      {
        rdyn_set_error("hipSetDevice(%d) failed", device);
        return RDYN_ERR_NO_DEVICE;
      }
      switched = true;
    }
    return RDYN_OK;
  }
  ~DeviceGuard()
  {
    if (switched) (void)hipSetDevice(prev);
  }
};

int device_const(const rdyn_chain* c, const RdynChainConst** out)
{
  if (c->long_chain())
  {
    rdyn_set_error("chains of more than %d joints are served through their reduced companion only", RDYN_MAX_SWEPT_JOINTS);
    return RDYN_ERR_UNSUPPORTED;
  }
  int dev = 0;
  RDYN_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(c->mu);
  auto it = c->dev_const.find(dev);
  if (it == c->dev_const.end())
  {
    RdynChainConst* d = nullptr;
    RDYN_HIP_TRY(hipMalloc((void**)&d, sizeof(RdynChainConst)));
    hipError_t e = hipMemcpy(d, &c->host_const, sizeof(RdynChainConst), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
      (void)hipFree(d);
      rdyn_set_error("HIP error: %s (upload of chain constants)", hipGetErrorString(e));
      return RDYN_ERR_HIP;
    }
    it = c->dev_const.emplace(dev, d).first;
  }
  *out = it->second;
  return RDYN_OK;
}

// constants of a chain of more than RDYN_MAX_SWEPT_JOINTS joints for the run-time-length kinematic kernels (rdyn_long_kin.hip)
int device_const_long(const rdyn_chain* c, const RdynLongChainConst** out)
{
  int dev = 0;
  RDYN_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(c->mu);
  auto it = c->dev_long.find(dev);
  if (it == c->dev_long.end())
  {
    RdynLongChainConst* d = nullptr;
    RDYN_HIP_TRY(hipMalloc((void**)&d, sizeof(RdynLongChainConst)));
    hipError_t e = hipMemcpy(d, &c->host_long, sizeof(RdynLongChainConst), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
      (void)hipFree(d);
      rdyn_set_error("HIP error: %s (upload of chain constants)", hipGetErrorString(e));
      return RDYN_ERR_HIP;
    }
    it = c->dev_long.emplace(dev, d).first;
  }
  *out = it->second;
  return RDYN_OK;
}

// device copy of the expansion blocks X_f of a chain with a reduced companion (rdyn_chain.hpp), current device
int device_expand(const rdyn_chain* c, const double** out)
{
  int dev = 0;
  RDYN_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(c->mu);
  auto it = c->dev_expand.find(dev);
  if (it == c->dev_expand.end())
  {
    double* d = nullptr;
    const size_t bytes = c->expand_X.size() * sizeof(double);
    RDYN_HIP_TRY(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemcpy(d, c->expand_X.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
      (void)hipFree(d);
      rdyn_set_error("HIP error: %s (upload of the expansion blocks)", hipGetErrorString(e));
      return RDYN_ERR_HIP;
    }
    it = c->dev_expand.emplace(dev, d).first;
  }
  *out = it->second;
  return RDYN_OK;
}

// temporary normal equations of the reduced chain at the end of a Gram workspace: G_red | c_red | bb_red
size_t reduce_tmp_bytes(int cols_red) { return (((size_t)cols_red * cols_red + cols_red + 1) * sizeof(double) + 255) & ~(size_t)255; }

// How an entry point serves a chain of more than RDYN_MAX_SWEPT_JOINTS joints: through its reduced companion (regressor, torque,
// inertia, normal equations, R factors, IK: at most RDYN_MAX_SWEPT_JOINTS input joints) or by the run-time-length kinematic kernels
// (frames, Jacobians, twists, wrenches: any number of input joints)
enum { LONG_NONE = 0, LONG_COMPANION = 1, LONG_KERNELS = 2 };
int check_batch(const rdyn_chain* c, const rdyn_batch* b, bool need_dq, bool need_ddq, const char* fn, int long_mode)
{
  if (!c || !b)
  {
    rdyn_set_error("%s: null chain or batch", fn);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (c->long_chain() && long_mode == LONG_NONE)
  {
    rdyn_set_error("%s: chains of more than %d joints are not served by this entry point", fn, RDYN_MAX_SWEPT_JOINTS);
    return RDYN_ERR_UNSUPPORTED;
  }
  if (c->long_chain() && long_mode == LONG_COMPANION && !c->reduced)
  {
    rdyn_set_error("%s: a chain of %d joints needs at most %d input joints", fn, c->n_joints(), RDYN_MAX_SWEPT_JOINTS);
    return RDYN_ERR_UNSUPPORTED;
  }
  if (b->n_samples < 0 || (b->layout != RDYN_LAYOUT_SAMPLE_MAJOR && b->layout != RDYN_LAYOUT_ELEMENT_MAJOR))
  {
    rdyn_set_error("%s: invalid n_samples or layout", fn);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (c->n_joints() < 1)
  {
    rdyn_set_error("%s: chain has no joints", fn);
    return RDYN_ERR_UNSUPPORTED;
  }
  if (b->n_samples > 0 && (!b->q || (need_dq && !b->dq) || (need_ddq && !b->ddq)))
  {
    // the reference checks only getRegressor's dimensions (primitives_impl.h:1299-1309)
    rdyn_set_error("Input data dimensions mismatch");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  return RDYN_OK;
}

// every non-null pointer starts on a 128-byte line (the staged copy-out of the sample-major records writes whole lines)
template <class... P>
inline bool lines_aligned(P... p)
{
  return (((uintptr_t)p | ...) & 127u) == 0;
}

// per-sample / per-element strides of a record of `elems` doubles
inline void rec_strides(const rdyn_batch* b, int64_t elems, int64_t* ss, int64_t* se)
{
  if (b->layout == RDYN_LAYOUT_SAMPLE_MAJOR)
  {
    *ss = elems;
    *se = 1;
  }
  else
  {
    *ss = 1;
    *se = b->n_samples;
  }
}

// Which row-contiguous regressor kernel (rdyn_image.hip) serves this layout: 0 none, 1 the per-sample image (stride_row 1, stride_col n,
// stride_sample >= n P), 2 the stacked (N n) x P matrix (stride_sample == n).  Needs the input joints in chain order and a compiled
// fixed-joint pattern (*fix_mask: bit f = chain joint f is not an input joint), and a 16-byte aligned Y: the copy-out moves 16-byte
// chunks whose addresses are Y + a multiple of 16 (an 8-byte aligned base -- a view at an odd double offset -- would put the last
// chunk of every image 8 bytes past its end); such calls keep the row-pair / strided kernels.
int image_route(const rdyn_chain* c, const rdyn_regressor_layout* yl, int64_t n_samples, const double* Y, bool multi, unsigned* fix_mask, bool* mapped = nullptr)
{
  if (mapped) *mapped = false;
  const int n = c->n_active(), nJ = c->n_joints();
  const bool lay_image = yl->stride_row == 1 && yl->stride_col == n && yl->stride_sample >= (int64_t)n * 10 * nJ;
  const bool lay_stacked = yl->stride_row == 1 && yl->stride_sample == n && yl->stride_col >= n_samples * n && !probe_env("RDYN_NO_STACKED_LDS");
  if (!(lay_image || lay_stacked) || probe_env("RDYN_NO_IMAGE")) return 0;
  // less than one wave of samples (the facade's single-sample getRegressor): the staging machinery costs more than it saves (19.5 us per
  // call at N = 1 against 11 for the kernels that store from the computing lane, profiles/r6/perf_sheet.txt) -- the row-pair / strided kernels
  if (n_samples < 64 && !multi) return 0;
  if (((uintptr_t)Y & 15u) != 0) return 0;
  if (lay_stacked && (yl->stride_col * 8) % 16 != 0) return 0;  // every column must start 16-byte aligned too
  unsigned fix = (nJ >= 32) ? 0u : ((1u << nJ) - 1u);
  bool ordered_inputs = true;
  for (int j = 0; j < n; ++j)
  {
    if (j > 0 && c->active[j] <= c->active[j - 1]) ordered_inputs = false;
    fix &= ~(1u << c->active[j]);
  }
  *fix_mask = fix;
  if (ordered_inputs && rdyn_image_supported(nJ, fix, yl->stride_sample, multi)) return lay_stacked ? 2 : 1;
  // input joints out of chain order, or joints that are not input joints somewhere else than the compiled head / tail patterns:
  // per-sample images go through the run-time row map (k_image_sweep<.., MAP>); everything else keeps the row-pair / strided kernels
  if (mapped && !multi && lay_image && !lay_stacked && (ordered_inputs || c->sorted) && rdyn_image_map_supported(nJ, fix, yl->stride_sample) &&
      !probe_env("RDYN_NO_IMAGE_MAP"))
  {
    *mapped = true;
    return 1;
  }
  return 0;
}

int run_local(const rdyn_chain* c, const rdyn_batch* b, int mode, double* tau, double* Y, const rdyn_regressor_layout* yl, double* M,
              bool use_dq, bool use_ddq)
{
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynSweepArgs a;
  memset(&a, 0, sizeof a);
  // a chain longer than the kernels sweep: its reduced companion has the same input joints, torques and joint-space inertia; the
  // dense regressor is the companion's with every body's ten-vector multiplied by the blocks of the links that ride on it
  const rdyn_chain* const full = c;
  if (c->long_chain())
  {
    if (mode == RDYN_MODE_REGRESSOR)
    {
      mode = RDYN_MODE_REGRESSOR_EXPAND;
      st = device_expand(full, &a.expand_X);
      if (st != RDYN_OK) return st;
      a.expand_n = full->n_joints();
      for (int f = 0; f < full->n_joints(); ++f) a.expand_red_of[f] = full->red_of[f];
    }
    c = full->reduced.get();
  }
  st = device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  const int n = c->n_active();
  a.q = b->q;
  a.dq = use_dq ? b->dq : nullptr;
  a.ddq = use_ddq ? b->ddq : nullptr;
  a.n_samples = b->n_samples;
  rec_strides(b, n, &a.in_ss, &a.in_sj);
  a.tau = tau;
  a.tau_ss = a.in_ss;
  a.tau_sj = a.in_sj;
  a.Y = Y;
  if (yl)
  {
    a.y_ss = yl->stride_sample;
    a.y_sr = yl->stride_row;
    a.y_sc = yl->stride_col;
  }
  if (mode == RDYN_MODE_REGRESSOR_EXPAND && yl && yl->stride_row == 1 && ((uintptr_t)Y & 15) == 0 && !probe_env("RDYN_NO_EXPAND_STAGING"))
  {
    // row-contiguous layouts: every link's block through the wave's LDS tile, copied out 16 bytes per lane (k_expand_staged)
    if (yl->stride_col == n && yl->stride_sample % 2 == 0) a.expand_stage = 1;        // per-sample images
    else if (yl->stride_sample == n && yl->stride_col % 2 == 0) a.expand_stage = 2;   // stacked (N n) x P matrix
    if (a.expand_stage) mode = RDYN_MODE_REGRESSOR_EXPAND_STAGED;
  }
  // per-sample images of a long chain whose companion has up to 6 joints: the image kernel (whole-line copy-out through the per-sample
  // ring) forms the blocks of the riding links one at a time (k_image_sweep<.., EXPAND>)
  bool expand_image = false;
  if (mode == RDYN_MODE_REGRESSOR_EXPAND_STAGED && a.expand_stage == 1 && c->n_active() == c->n_joints() && !probe_env("RDYN_NO_EXPAND_IMAGE") &&
      rdyn_image_expand_supported(c->n_joints(), full->n_joints(), yl->stride_sample))
  {
    const int nr = c->n_joints(), nf = full->n_joints();
    bool mono = true;
    for (int g = 1; g < nf; ++g) mono = mono && full->red_of[g] >= full->red_of[g - 1];
    if (mono)
    {
      for (int f = 0; f <= nr; ++f)
      {
        int g = 0;
        while (g < nf && full->red_of[g] < f) ++g;
        a.expand_first[f] = g;   // (f = nr: nf)
      }
      for (int i = 0; i < n; ++i) a.row_map[c->active[i]] = i;  // reduced joint -> the caller's input index
      expand_image = true;
    }
  }
  if (yl && (mode == RDYN_MODE_REGRESSOR || mode == RDYN_MODE_REGRESSOR_EXPAND || mode == RDYN_MODE_REGRESSOR_EXPAND_STAGED))
  {
    // the one-thread-per-sample kernel addresses a workgroup's 256 samples with 32-bit lane offsets: free strides must be
    // positive and keep 255 * stride_sample * 8 inside 32 bits (the presets of rdyn.h are far below that)
    if (yl->stride_sample < 1 || yl->stride_row < 1 || yl->stride_col < 1 || yl->stride_sample > (int64_t)0xFFFFFFFFll / (8 * 255))
    {
      rdyn_set_error("rdyn_regressor: strides must be positive and stride_sample below %lld doubles", (long long)((int64_t)0xFFFFFFFFll / (8 * 255)));
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  }
  a.M = M;
  rec_strides(b, (int64_t)n * n, &a.m_ss, &a.m_se);
  // torque / inertia records in the drop-in layout: through the wave's LDS tile, whole lines (rdyn_record_stage.h)
  if ((mode == RDYN_MODE_TORQUE || mode == RDYN_MODE_INERTIA) && b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(tau, M) &&
      !probe_env("RDYN_NO_RECORD_STAGING"))
    a.staged = mode == RDYN_MODE_INERTIA ? n * n : n;
  // Row-contiguous regressor layouts (stacked column-major, per-sample Eigen image) with sample-major inputs:
  // ceil(n/2) lanes per sample, 16-byte row-pair stores (k_rowpair_sweep) instead of 8-byte strided stores.
  const bool rowpair = mode == RDYN_MODE_REGRESSOR && yl && yl->stride_row == 1 && n >= 2 && n <= 10 &&
                       b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && b->n_samples * ((n + 1) / 2) < (int64_t)0xFFFFFF00ll &&
                       !probe_env("RDYN_NO_ROWPAIR");
  // the drop-in per-sample image (either input layout): one thread per sample, link blocks staged through LDS (rdyn_image.hip)
  // and the stacked column-major (N n) x P matrix (stride_sample == n): same kernel, column-major staging tile per link
  unsigned fix_mask = 0;
  bool mapped = false;
  const bool image = mode == RDYN_MODE_REGRESSOR && yl && image_route(c, yl, b->n_samples, Y, false, &fix_mask, &mapped) != 0;
  if (expand_image)
  {
    // (the chunk permutation of the per-sample images, see below)
    const unsigned grid = (unsigned)((b->n_samples + 63) / 64);
    auto gcd = [](unsigned x, unsigned y) { while (y) { const unsigned t = x % y; x = y; y = t; } return x; };
    unsigned m = 257;
    while (gcd(m, grid) != 1) ++m;
    a.blk_mul = (grid > 4 * m && !probe_env("RDYN_NO_IMAGE_SCATTER")) ? m : 0;
    RDYN_HIP_TRY(rdyn_launch_image_sweep(c->n_joints(), 0u, a, (hipStream_t)b->stream, 2));
  }
  else if (image)
  {
    if (mapped)
    {
      // the swept chain: the sorted view when the input joints were listed out of chain order (same joints, rows in chain order)
      const rdyn_chain* const so = c->sorted ? c->sorted.get() : c;
      st = device_const(so, &a.chain);
      if (st != RDYN_OK) return st;
      for (int f = 0; f < RDYN_MAX_SWEPT_JOINTS; ++f) a.row_map[f] = -1;
      for (int r = 0; r < n; ++r) a.row_map[so->active[r]] = r < (int)so->row_input.size() ? so->row_input[r] : r;
    }
    // per-sample images: the workgroups visit the 64-sample chunks of the batch in a PERMUTED order (chunk = b * 257 mod grid), so that
    // the ~2 000 waves resident at a time write all over the output instead of one contiguous ~370 MB window of it.  Measured (round 6,
    // tools/probe_scatter.py, profiles/r6/probe_scatter.txt: ten 2.88 GB output allocations per process): 516-523 -> 450-455 us per 1e6
    // evaluations in half of the allocations, 537-547 -> 508-520 in most others, never slower; multipliers 17 .. 4097 alike.  The stacked
    // matrix (60 column fronts already) loses 2-5 % with it and keeps the workgroups in order.
    if (image_route(c, yl, b->n_samples, Y, false, &fix_mask, &mapped) == 1 && !probe_env("RDYN_NO_IMAGE_SCATTER"))
    {
      const unsigned grid = (unsigned)((b->n_samples + 63) / 64);
      auto gcd = [](unsigned x, unsigned y) { while (y) { const unsigned t = x % y; x = y; y = t; } return x; };
      unsigned m = 257;
      while (gcd(m, grid) != 1) ++m;
      a.blk_mul = grid > 4 * m ? m : 0;
    }
    RDYN_HIP_TRY(rdyn_launch_image_sweep(c->n_joints(), fix_mask, a, (hipStream_t)b->stream, mapped));
  }
  else if (rowpair)
  {
    // the kernel addresses Y with a 32-bit per-lane byte offset: split so that every launch spans < 4 GB of Y
    const int64_t span = (a.y_ss > 0 ? a.y_ss : 1) * 8;
    int64_t chunk = ((int64_t)0xF0000000ll / span) & ~(int64_t)255;
    if (chunk < 256) chunk = 256;
    for (int64_t s0 = 0; s0 < b->n_samples; s0 += chunk)
    {
      RdynSweepArgs p = a;
      p.n_samples = (b->n_samples - s0 < chunk) ? b->n_samples - s0 : chunk;
      p.q = a.q + s0 * a.in_ss;
      p.dq = a.dq + s0 * a.in_ss;
      p.ddq = a.ddq + s0 * a.in_ss;
      p.Y = a.Y + s0 * a.y_ss;
      if (a.tau) p.tau = a.tau + s0 * a.tau_ss;
      RDYN_HIP_TRY(rdyn_launch_rowpair_sweep(c->n_joints(), n, p, (hipStream_t)b->stream));
    }
  }
  else
    RDYN_HIP_TRY(rdyn_launch_local_sweep(c->n_joints(), mode, a, (hipStream_t)b->stream));
  return RDYN_OK;
}

// regressor / inertia of a chain with more input joints than the unrolled kernels sweep (no reduced companion): the run-time-length
// kernels of rdyn_long_local.hip -- rolled link and row loops, per-joint state in wave-private LDS
int run_long_local(const rdyn_chain* c, const rdyn_batch* b, int mode, double* tau, double* Y, const rdyn_regressor_layout* yl, double* M)
{
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynLongLocalArgs a;
  memset(&a, 0, sizeof a);
  st = device_const_long(c, &a.chain_long);
  if (st != RDYN_OK) return st;
  const int n = c->n_active();
  a.q = b->q;
  a.dq = b->dq;
  a.ddq = b->ddq;
  a.n_samples = b->n_samples;
  rec_strides(b, n, &a.in_ss, &a.in_sj);
  a.tau = tau;
  a.tau_ss = a.in_ss;
  a.tau_sj = a.in_sj;
  a.Y = Y;
  if (yl)
  {
    if (yl->stride_sample < 1 || yl->stride_row < 1 || yl->stride_col < 1)
    {
      rdyn_set_error("rdyn_regressor: strides must be positive");
      return RDYN_ERR_INVALID_ARGUMENT;
    }
    a.y_ss = yl->stride_sample;
    a.y_sr = yl->stride_row;
    a.y_sc = yl->stride_col;
    // row-contiguous layouts: every link's block through the wave's LDS tile, copied out 16 bytes per lane
    if (yl->stride_row == 1 && ((uintptr_t)Y & 15) == 0 && !probe_env("RDYN_NO_EXPAND_STAGING"))
    {
      if (yl->stride_col == n && yl->stride_sample % 2 == 0 && yl->stride_sample >= (int64_t)n * 10 * c->n_joints()) a.stage = 1;
      else if (yl->stride_sample == n && yl->stride_col % 2 == 0 && yl->stride_col >= b->n_samples * n) a.stage = 2;
    }
  }
  a.n_active = n;
  a.M = M;
  rec_strides(b, (int64_t)n * n, &a.m_ss, &a.m_se);
  RDYN_HIP_TRY(rdyn_launch_long_local(mode, c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

}  // namespace

extern "C"
{

// a chain of more than RDYN_MAX_SWEPT_JOINTS joints whose input joints are too many for the companion: the joint torques read off the
// wrench recursion of the run-time-length kernels (primitives_impl.h:1264-1272; use_ddq = false: DDq = 0)
static int long_chain_torque(const rdyn_chain* c, const rdyn_batch* b, double* tau, bool use_ddq)
{
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynKinExtArgs e;
  memset(&e, 0, sizeof e);
  st = device_const_long(c, &e.chain_long);
  if (st != RDYN_OK) return st;
  e.q = b->q;
  e.dq = b->dq;
  e.ddq = use_ddq ? b->ddq : nullptr;
  e.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &e.in_ss, &e.in_sj);
  e.tau = tau;
  e.tau_ss = e.in_ss;
  e.tau_sj = e.in_sj;
  e.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(tau) && !probe_env("RDYN_NO_RECORD_STAGING");
  RDYN_HIP_TRY(rdyn_launch_long_ext(c->n_joints(), e, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_joint_torque(const rdyn_chain* c, const rdyn_batch* b, double* tau)
{
  const bool by_wrench = c && c->long_chain() && !c->reduced;
  int st = check_batch(c, b, true, true, "rdyn_joint_torque", by_wrench ? LONG_KERNELS : LONG_COMPANION);
  if (st != RDYN_OK) return st;
  if (!tau && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_joint_torque: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (by_wrench) return long_chain_torque(c, b, tau, true);
  return run_local(c, b, RDYN_MODE_TORQUE, tau, nullptr, nullptr, nullptr, true, true);
}

int rdyn_joint_torque_nonlinear(const rdyn_chain* c, const rdyn_batch* b, double* tau)
{
  const bool by_wrench = c && c->long_chain() && !c->reduced;
  int st = check_batch(c, b, true, false, "rdyn_joint_torque_nonlinear", by_wrench ? LONG_KERNELS : LONG_COMPANION);
  if (st != RDYN_OK) return st;
  if (!tau && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_joint_torque_nonlinear: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (by_wrench) return long_chain_torque(c, b, tau, false);
  return run_local(c, b, RDYN_MODE_TORQUE, tau, nullptr, nullptr, nullptr, true, false);  // DDq = 0, primitives_impl.h:1287-1288
}

int rdyn_regressor(const rdyn_chain* c, const rdyn_batch* b, double* tau, double* Y, const rdyn_regressor_layout* yl)
{
  int st = check_batch(c, b, true, true, "rdyn_regressor", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (b->n_samples > 0 && (!Y || !yl))
  {
    rdyn_set_error("rdyn_regressor: null output or layout");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (c->long_chain() && !c->reduced) return run_long_local(c, b, RDYN_MODE_REGRESSOR, tau, Y, yl, nullptr);
  return run_local(c, b, RDYN_MODE_REGRESSOR, tau, Y, yl, nullptr, true, true);
}

int rdyn_joint_inertia(const rdyn_chain* c, const rdyn_batch* b, double* M)
{
  int st = check_batch(c, b, false, false, "rdyn_joint_inertia", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!M && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_joint_inertia: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (c->long_chain() && !c->reduced) return run_long_local(c, b, RDYN_MODE_INERTIA, nullptr, nullptr, nullptr, M);
  return run_local(c, b, RDYN_MODE_INERTIA, nullptr, nullptr, nullptr, M, false, false);
}

static int run_base(const rdyn_chain* c, const rdyn_batch* b, double* T_bt, double* T_links, double* J, double* tw, double* dtw,
                    int j_link = -1)
{
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynKinArgs a;
  memset(&a, 0, sizeof a);
  st = c->long_chain() ? device_const_long(c, &a.chain_long) : device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  const int n = c->n_active(), L = c->n_joints() + 1;
  a.q = b->q;
  a.dq = (tw || dtw) ? b->dq : nullptr;
  a.ddq = dtw ? b->ddq : nullptr;
  a.n_samples = b->n_samples;
  rec_strides(b, n, &a.in_ss, &a.in_sj);
  int64_t se;
  a.T_bt = T_bt;
  rec_strides(b, 12, &a.tb_ss, &se);
  a.T_links = T_links;
  rec_strides(b, 12 * (int64_t)L, &a.tl_ss, &se);
  a.J = J;
  a.j_link = j_link < 0 ? c->n_joints() : j_link;
  rec_strides(b, 6 * (int64_t)n, &a.j_ss, &se);
  a.twists = tw;
  a.dtwists = dtw;
  rec_strides(b, 6 * (int64_t)L, &a.tw_ss, &se);
  a.out_se = se;
  a.n_active = n;
  // the drop-in layout: records leave through wave-private LDS in whole lines (rdyn_record_stage.h) when every output starts on a line
  a.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(T_bt, T_links, J, tw, dtw) && !probe_env("RDYN_NO_RECORD_STAGING");
  if (c->long_chain())
  {
    for (int l = 0; l < a.j_link; ++l) a.j_up += c->host_joints[l].in_idx >= 0 ? 1 : 0;
    RDYN_HIP_TRY(rdyn_launch_long_base(a, (hipStream_t)b->stream));
  }
  else
    RDYN_HIP_TRY(rdyn_launch_base_sweep(c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

// ---- batched local inverse kinematics -------------------------------------------------------------------------
int rdyn_local_ik(const rdyn_chain* c, const rdyn_batch* b, const double* T_target, const double* weight, double toll,
                  int max_iterations, double* sol, int32_t* status, int32_t* iterations)
{
  return rdyn_local_ik_damped(c, b, T_target, weight, toll, 0.0, max_iterations, sol, status, iterations);
}

int rdyn_local_ik_damped(const rdyn_chain* c, const rdyn_batch* b, const double* T_target, const double* weight, double toll,
                         double damping, int max_iterations, double* sol, int32_t* status, int32_t* iterations)
{
  int st = check_batch(c, b, false, false, "rdyn_local_ik", LONG_COMPANION);  // batch->q = the seeds
  if (st != RDYN_OK) return st;
  if (b->n_samples > 0 && (!T_target || !sol))
  {
    rdyn_set_error("rdyn_local_ik: null target or solution pointer");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (max_iterations < 0 || !(toll >= 0.0) || !(damping >= 0.0))
  {
    rdyn_set_error("rdyn_local_ik: negative iteration cap, tolerance or damping");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynIkArgs a;
  memset(&a, 0, sizeof a);
  if (c->long_chain())
  {
    // the pose and the Jacobian of the tool need only the input joints: the reduced companion (the fixed frames folded into the
    // joint origins) is iterated, its tool frame followed by the constant frames behind the last input joint
    a.has_tail = 1;
    memcpy(a.tail_R, c->tail_R, sizeof a.tail_R);
    memcpy(a.tail_t, c->tail_t, sizeof a.tail_t);
    c = c->reduced.get();
  }
  st = device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  a.T_target = T_target;
  rec_strides(b, 12, &a.tt_ss, &a.tt_se);
  a.seed = b->q;
  a.sol = sol;
  a.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &a.in_ss, &a.in_sj);
  for (int i = 0; i < 6; ++i) a.weight[i] = weight ? weight[i] : 1.0;
  for (int j = 0; j < c->n_joints(); ++j)
  {
    a.q_min[j] = c->q_min[j];
    a.q_max[j] = c->q_max[j];
  }
  a.toll = toll;
  a.damping = damping;
  a.max_iter = max_iterations;
  a.status = status;
  a.iterations = iterations;
  // Staged launches when the per-pose results are available to carry the state: most poses settle within a few updates,
  // the rest would keep every wave alive for the whole cap.  The first launch runs everyone for 8 updates; every further
  // launch gathers the poses still running into dense waves and continues only those up to the next boundary (same
  // arithmetic per pose, rdyn_ik.hip: k_local_ik_resume).
  // (boundaries at 4 and 16 as well were measured slower: re-packing 42 % of the poses after 4 updates costs more than the
  // idle lanes it saves -- cap 8: 1.64 ms vs 1.11 ms)
  static const int kIkStages[] = {8};  // update counts after which the survivors are re-packed
  const int n_stages = (int)(sizeof kIkStages / sizeof kIkStages[0]);
  if (status && iterations)
  {
    int done = 0;  // updates every still-running pose has performed so far
    for (int k = 0; k <= n_stages; ++k)
    {
      const int upto = (k < n_stages && kIkStages[k] < max_iterations) ? kIkStages[k] : max_iterations;
      a.it_stage = done;  // 0: first launch (all poses, from the seeds); > 0: resume the poses stopped at `done` updates
      a.max_iter = upto;
      RDYN_HIP_TRY(rdyn_launch_local_ik(c->n_joints(), a, (hipStream_t)b->stream));
      done = upto;
      if (upto == max_iterations) break;
    }
    return RDYN_OK;
  }
  RDYN_HIP_TRY(rdyn_launch_local_ik(c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_frame_distance(int64_t n_pairs, const double* T_wa, const double* T_wb, int layout, int kind, double* distance, double* jacobian,
                        int device, void* stream)
{
  if (n_pairs < 0 || (layout != RDYN_LAYOUT_SAMPLE_MAJOR && layout != RDYN_LAYOUT_ELEMENT_MAJOR) || kind < 0 || kind > 2 ||
      (n_pairs > 0 && (!T_wa || !T_wb || !distance)) || (jacobian && kind != RDYN_FRAME_DISTANCE_QUAT_JAC))
  {
    rdyn_set_error("rdyn_frame_distance: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (n_pairs == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(device);
  if (st != RDYN_OK) return st;
  RdynFrameDistanceArgs a;
  memset(&a, 0, sizeof a);
  rdyn_batch b;
  memset(&b, 0, sizeof b);
  b.n_samples = n_pairs;
  b.layout = layout;
  a.T_wa = T_wa;
  a.T_wb = T_wb;
  a.n = n_pairs;
  a.kind = kind;
  a.distance = distance;
  a.jacobian = jacobian;
  rec_strides(&b, 12, &a.t_ss, &a.t_se);
  rec_strides(&b, 6, &a.d_ss, &a.d_se);
  rec_strides(&b, 36, &a.j_ss, &a.j_se);
  RDYN_HIP_TRY(rdyn_launch_frame_distance(a, (hipStream_t)stream));
  return RDYN_OK;
}

// ---- split / jerk sweeps, external wrenches ---------------------------------------------------------------
int rdyn_twist_parts(const rdyn_chain* c, const rdyn_batch* b, const double* dddq, double* dtw_lin, double* dtw_nonlin, double* ddtw)
{
  int st = check_batch(c, b, dtw_nonlin || ddtw, dtw_lin || ddtw, "rdyn_twist_parts", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if ((!dtw_lin && !dtw_nonlin && !ddtw) || (ddtw && !dddq && b->n_samples > 0))
  {
    rdyn_set_error("rdyn_twist_parts: no output, or ddtwists without dddq");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynKinExtArgs a;
  memset(&a, 0, sizeof a);
  st = c->long_chain() ? device_const_long(c, &a.chain_long) : device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  a.q = b->q;
  a.dq = b->dq;
  a.ddq = b->ddq;
  a.dddq = dddq;
  a.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &a.in_ss, &a.in_sj);
  rec_strides(b, 6 * (int64_t)(c->n_joints() + 1), &a.out_ss, &a.out_se);
  a.dtw_lin = dtw_lin;
  a.dtw_nonlin = dtw_nonlin;
  a.ddtw = ddtw;
  a.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(dtw_lin, dtw_nonlin, ddtw) && !probe_env("RDYN_NO_RECORD_STAGING");
  if (c->long_chain())
    RDYN_HIP_TRY(rdyn_launch_long_ext(c->n_joints(), a, (hipStream_t)b->stream));
  else
    RDYN_HIP_TRY(rdyn_launch_base_ext(c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_jerk_parts(const rdyn_chain* c, const rdyn_batch* b, const double* dddq, double* ddtw_lin, double* ddtw_nonlin)
{
  int st = check_batch(c, b, ddtw_nonlin != nullptr, ddtw_nonlin != nullptr, "rdyn_jerk_parts", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if ((!ddtw_lin && !ddtw_nonlin) || (ddtw_lin && !dddq && b->n_samples > 0))
  {
    rdyn_set_error("rdyn_jerk_parts: no output, or the linear part without dddq");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynKinExtArgs a;
  memset(&a, 0, sizeof a);
  st = c->long_chain() ? device_const_long(c, &a.chain_long) : device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  a.q = b->q;
  a.dq = ddtw_nonlin ? b->dq : nullptr;
  a.ddq = ddtw_nonlin ? b->ddq : nullptr;
  a.dddq = ddtw_lin ? dddq : nullptr;
  a.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &a.in_ss, &a.in_sj);
  rec_strides(b, 6 * (int64_t)(c->n_joints() + 1), &a.out_ss, &a.out_se);
  a.ddtw_lin = ddtw_lin;
  a.ddtw_nonlin = ddtw_nonlin;
  a.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(ddtw_lin, ddtw_nonlin) && !probe_env("RDYN_NO_RECORD_STAGING");
  if (c->long_chain())
    RDYN_HIP_TRY(rdyn_launch_long_ext(c->n_joints(), a, (hipStream_t)b->stream));
  else
    RDYN_HIP_TRY(rdyn_launch_base_ext(c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_wrench(const rdyn_chain* c, const rdyn_batch* b, const double* ext, double* wrenches)
{
  int st = check_batch(c, b, true, true, "rdyn_wrench", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!wrenches && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_wrench: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  RdynKinExtArgs a;
  memset(&a, 0, sizeof a);
  st = c->long_chain() ? device_const_long(c, &a.chain_long) : device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  a.q = b->q;
  a.dq = b->dq;
  a.ddq = b->ddq;
  a.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &a.in_ss, &a.in_sj);
  rec_strides(b, 6 * (int64_t)(c->n_joints() + 1), &a.out_ss, &a.out_se);
  a.wrench = wrenches;
  a.ext = ext;
  a.ext_ss = a.out_ss;
  a.ext_se = a.out_se;
  a.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(wrenches) && !probe_env("RDYN_NO_RECORD_STAGING");
  a.ext_staged = a.staged && ext && ((uintptr_t)ext & 15u) == 0 && !probe_env("RDYN_NO_EXT_STAGING");
  if (c->long_chain())
    RDYN_HIP_TRY(rdyn_launch_long_ext(c->n_joints(), a, (hipStream_t)b->stream));
  else
    RDYN_HIP_TRY(rdyn_launch_base_ext(c->n_joints(), a, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_joint_torque_ext(const rdyn_chain* c, const rdyn_batch* b, const double* ext, double* tau)
{
  int st = check_batch(c, b, true, true, "rdyn_joint_torque_ext", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if ((!tau || !ext) && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_joint_torque_ext: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  if (c->long_chain())
  {
    // the wrench recursion of the run-time-length kernels, the joint torques read off the link wrenches (primitives_impl.h:1264-1272)
    RdynKinExtArgs e;
    memset(&e, 0, sizeof e);
    st = device_const_long(c, &e.chain_long);
    if (st != RDYN_OK) return st;
    e.q = b->q;
    e.dq = b->dq;
    e.ddq = b->ddq;
    e.n_samples = b->n_samples;
    rec_strides(b, c->n_active(), &e.in_ss, &e.in_sj);
    e.tau = tau;
    e.tau_ss = e.in_ss;
    e.tau_sj = e.in_sj;
    e.ext = ext;
    rec_strides(b, 6 * (int64_t)(c->n_joints() + 1), &e.ext_ss, &e.ext_se);
    e.staged = b->layout == RDYN_LAYOUT_SAMPLE_MAJOR && lines_aligned(tau) && !probe_env("RDYN_NO_RECORD_STAGING");
    RDYN_HIP_TRY(rdyn_launch_long_ext(c->n_joints(), e, (hipStream_t)b->stream));
    return RDYN_OK;
  }
  RdynSweepArgs a;
  memset(&a, 0, sizeof a);
  st = device_const(c, &a.chain);
  if (st != RDYN_OK) return st;
  a.q = b->q;
  a.dq = b->dq;
  a.ddq = b->ddq;
  a.n_samples = b->n_samples;
  rec_strides(b, c->n_active(), &a.in_ss, &a.in_sj);
  a.tau = tau;
  a.tau_ss = a.in_ss;
  a.tau_sj = a.in_sj;
  a.ext = ext;
  rec_strides(b, 6 * (int64_t)(c->n_joints() + 1), &a.ext_ss, &a.ext_se);
  RDYN_HIP_TRY(rdyn_launch_local_sweep(c->n_joints(), RDYN_MODE_TORQUE, a, (hipStream_t)b->stream));
  return RDYN_OK;
}

// ---- additive components ---------------------------------------------------------------------------------
int rdyn_components_columns(const rdyn_component* comps, int n_comps)
{
  if (!comps || n_comps < 0) return -1;
  int k = 0;
  for (int i = 0; i < n_comps; ++i) k += (comps[i].type == RDYN_COMP_FRICTION2) ? 3 : 2;
  return k;
}

// validates the components and copies them with the constructor rules of the reference applied
static int fill_components(const rdyn_component* comps, int n_comps, int n_active, RdynComponentArgs* a)
{
  for (int i = 0; i < n_comps; ++i)
  {
    const rdyn_component& c = comps[i];
    if (c.type < RDYN_COMP_FRICTION1 || c.type > RDYN_COMP_SPRING || c.joint < 0 || c.joint >= n_active)
    {
      rdyn_set_error("component %d has an invalid type or joint", i);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
    a->comps[i].type = c.type;
    a->comps[i].joint = c.joint;
    a->comps[i].min_velocity = c.min_velocity < 1e-6 ? 1e-6 : c.min_velocity;  // friction_polynomial1.h:73-78
    a->comps[i].max_velocity = c.max_velocity <= 0 ? 1.0e6 : c.max_velocity;   // friction_polynomial1.h:81-86
    for (int k = 0; k < 3; ++k) a->comps[i].parameters[k] = c.parameters[k];
  }
  return RDYN_OK;
}

int rdyn_components_regressor(const rdyn_component* comps, int n_comps, int n_active, const rdyn_batch* b, double* C,
                              const rdyn_regressor_layout* cl, double* tau_add)
{
  if (!comps || n_comps < 1 || n_comps > RDYN_MAX_COMPONENTS || n_active < 1 || !b || (!C && !tau_add) || (C && !cl))
  {
    rdyn_set_error("rdyn_components_regressor: invalid argument (1..%d components, an output, a layout)", RDYN_MAX_COMPONENTS);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples < 0 || (b->layout != RDYN_LAYOUT_SAMPLE_MAJOR && b->layout != RDYN_LAYOUT_ELEMENT_MAJOR) ||
      (b->n_samples > 0 && (!b->q || !b->dq)))
  {
    rdyn_set_error("Input data dimensions mismatch");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  RdynComponentArgs a;
  memset(&a, 0, sizeof a);
  int cst = fill_components(comps, n_comps, n_active, &a);
  if (cst != RDYN_OK) return cst;
  if (b->n_samples == 0) return RDYN_OK;
  DeviceGuard g;
  int st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  a.q = b->q;
  a.dq = b->dq;
  a.n_samples = b->n_samples;
  rec_strides(b, n_active, &a.in_ss, &a.in_sj);
  a.n_active = n_active;
  a.n_comps = n_comps;
  a.C = C;
  if (cl)
  {
    a.c_ss = cl->stride_sample;
    a.c_sr = cl->stride_row;
    a.c_sc = cl->stride_col;
  }
  a.tau = tau_add;
  RDYN_HIP_TRY(rdyn_launch_components(a, (hipStream_t)b->stream));
  return RDYN_OK;
}

// ---- mixed-chain batch ----------------------------------------------------------------------------------
struct rdyn_multi_plan
{
  int device = 0;
  struct Group
  {
    int n_joints = 0;
    unsigned fix_mask = 0;  // image / stacked groups only: chain joints that are not input joints
    int kind = 0;      // 0: one thread per sample, strided stores (any layout); 1: per-sample images, 2: stacked matrices (rdyn_image.hip)
    int n_items = 0;
    int64_t max_samples = 0;
    RdynSweepArgs* table = nullptr;  // device
  };
  std::vector<Group> groups;
  ~rdyn_multi_plan()  // also runs when creation fails half-way
  {
    DeviceGuard g;
    if (g.enter(device) == RDYN_OK)
      for (auto& grp : groups) (void)hipFree(grp.table);
  }
};

int rdyn_multi_plan_create(const rdyn_multi_item* items, int n_items, rdyn_multi_plan** out)
{
  if (!items || n_items < 1 || !out)
  {
    rdyn_set_error("rdyn_multi_plan_create: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  DeviceGuard g;
  int st = g.enter(items[0].batch.device);
  if (st != RDYN_OK) return st;
  std::unique_ptr<rdyn_multi_plan> plan(new rdyn_multi_plan());
  RDYN_HIP_TRY(hipGetDevice(&plan->device));
  std::map<int, std::vector<RdynSweepArgs>> by_nj;
  std::map<int, int64_t> max_n;
  for (int i = 0; i < n_items; ++i)
  {
    const rdyn_multi_item& it = items[i];
    st = check_batch(it.chain, &it.batch, true, true, "rdyn_multi_plan_create", LONG_NONE);
    if (st != RDYN_OK) return st;
    if (it.batch.n_samples > 0 && !it.Y)
    {
      rdyn_set_error("rdyn_multi_plan_create: item %d has a null regressor output", i);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
    if (it.batch.n_samples > 0 &&  // an empty item's strides are never used (its row stride is its sample count: 0)
        (it.y_layout.stride_sample < 1 || it.y_layout.stride_row < 1 || it.y_layout.stride_col < 1 ||
         it.y_layout.stride_sample > (int64_t)0xFFFFFFFFll / (8 * 255)))
    {
      rdyn_set_error("rdyn_multi_plan_create: item %d: strides must be positive and stride_sample below %lld doubles", i,
                     (long long)((int64_t)0xFFFFFFFFll / (8 * 255)));
      return RDYN_ERR_INVALID_ARGUMENT;
    }
    RdynSweepArgs a;
    memset(&a, 0, sizeof a);
    st = device_const(it.chain, &a.chain);
    if (st != RDYN_OK) return st;
    a.q = it.batch.q;
    a.dq = it.batch.dq;
    a.ddq = it.batch.ddq;
    a.n_samples = it.batch.n_samples;
    rec_strides(&it.batch, it.chain->n_active(), &a.in_ss, &a.in_sj);
    a.tau = it.tau;
    a.tau_ss = a.in_ss;
    a.tau_sj = a.in_sj;
    a.Y = it.Y;
    a.y_ss = it.y_layout.stride_sample;
    a.y_sr = it.y_layout.stride_row;
    a.y_sc = it.y_layout.stride_col;
    const int nj = it.chain->n_joints();
    // one launch per (chain joints, fixed-joint pattern, kernel kind): row-contiguous layouts of chains of up to 8 input joints go
    // through the LDS-staged kernels (whole-line stores), the rest keeps the strided kernel.  Key = nj | mask << 8 | kind << 24.
    unsigned fix_mask = 0;
    const int kind = it.batch.n_samples > 0 ? image_route(it.chain, &it.y_layout, it.batch.n_samples, it.Y, true, &fix_mask) : 0;
    const int key = nj | ((kind ? (int)fix_mask : 0) << 8) | (kind << 24);
    by_nj[key].push_back(a);
    if (a.n_samples > max_n[key]) max_n[key] = a.n_samples;
  }
  for (auto& kv : by_nj)
  {
    rdyn_multi_plan::Group grp;
    grp.n_joints = kv.first & 0xFF;
    grp.fix_mask = (unsigned)(kv.first >> 8) & 0xFFFFu;
    grp.kind = kv.first >> 24;
    grp.n_items = (int)kv.second.size();
    grp.max_samples = max_n[kv.first];
    RDYN_HIP_TRY(hipMalloc((void**)&grp.table, sizeof(RdynSweepArgs) * kv.second.size()));
    plan->groups.push_back(grp);
    RDYN_HIP_TRY(hipMemcpy(grp.table, kv.second.data(), sizeof(RdynSweepArgs) * kv.second.size(), hipMemcpyHostToDevice));
  }
  *out = plan.release();
  return RDYN_OK;
}

int rdyn_multi_plan_regressor(const rdyn_multi_plan* plan, void* stream)
{
  if (!plan)
  {
    rdyn_set_error("rdyn_multi_plan_regressor: null plan");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g;
  int st = g.enter(plan->device);
  if (st != RDYN_OK) return st;
  for (const auto& grp : plan->groups)
  {
    if (grp.kind)
      RDYN_HIP_TRY(rdyn_launch_image_sweep_multi(grp.n_joints, grp.fix_mask, grp.kind == 2, grp.table, grp.n_items, grp.max_samples, (hipStream_t)stream));
    else
      RDYN_HIP_TRY(rdyn_launch_local_sweep_multi(grp.n_joints, RDYN_MODE_REGRESSOR, grp.table, grp.n_items, grp.max_samples, (hipStream_t)stream));
  }
  return RDYN_OK;
}

void rdyn_multi_plan_destroy(rdyn_multi_plan* plan)
{
  delete plan;  // the destructor releases the device tables
}

// ---- normal equations -----------------------------------------------------------------------------------
static const int kGramBlocks = 512;  // 2 workgroups per CU; each owns one slab of the workspace

static size_t gram_slab_bytes(int n_cols)
{
  const int nb = rdyn_gram_blocks_for(n_cols);
  return (size_t)kGramBlocks * (size_t)(nb * (nb + 1) / 2) * 256 * sizeof(double);
}

size_t rdyn_gram_workspace_bytes(int n_cols) { return (n_cols < 1 || rdyn_gram_blocks_for(n_cols) > 7) ? 0 : gram_slab_bytes(n_cols); }

static int gram_launch(const double* A, int64_t rows, int64_t lda, int n_cols, const double* bvec, double* G, double* cvec, double* bb,
                       int slab_accumulate, bool finish, int add_to_output, void* workspace, hipStream_t st,
                       int64_t row_block = 0, const int* first_col = nullptr, int n_row_blocks = 0)
{
  RdynGramArgs a;
  memset(&a, 0, sizeof a);
  a.row_block = first_col ? row_block : 0;
  for (int j = 0; first_col && j < n_row_blocks && j < RDYN_MAX_SWEPT_JOINTS; ++j) a.first_col[j] = first_col[j];
  a.A = A;
  a.b = bvec;
  a.rows = rows;
  a.lda = lda;
  a.P = n_cols;
  a.accumulate = slab_accumulate;
  a.add_to_output = add_to_output;
  a.slabs = (double*)workspace;
  a.G = G;
  a.c = cvec;
  a.bb = bb;
  RDYN_HIP_TRY(rdyn_launch_gram(a, kGramBlocks, st));
  if (finish) RDYN_HIP_TRY(rdyn_launch_gram_finish(a, kGramBlocks, st));
  return RDYN_OK;
}

int rdyn_gram(const double* A, int64_t rows, int64_t lda, int n_cols, const double* bvec, double* G, double* cvec, double* bb,
              int accumulate, void* workspace, size_t workspace_bytes, int device, void* stream)
{
  if (!A || !G || rows < 0 || lda < rows || n_cols < 1 || !workspace)
  {
    rdyn_set_error("rdyn_gram: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (rdyn_gram_blocks_for(n_cols) > 7)
  {
    rdyn_set_error("rdyn_gram: at most 111 columns are supported");
    return RDYN_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < gram_slab_bytes(n_cols))
  {
    rdyn_set_error("rdyn_gram: workspace too small (%zu < %zu bytes)", workspace_bytes, gram_slab_bytes(n_cols));
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g;
  int st = g.enter(device);
  if (st != RDYN_OK) return st;
  return gram_launch(A, rows, lda, n_cols, bvec, G, cvec, bb, 0, true, accumulate ? 1 : 0, workspace, (hipStream_t)stream);
}

static int64_t default_chunk(int64_t chunk) { return chunk > 0 ? chunk : 32768; }
// Fused regressor->Gram kernel: persistent workgroups, each with its own 256-sample tile image (rewritten per tile, so
// it stays in L2 / Infinity Cache): 256 workgroups x 256 samples x n x (P + 1) doubles = 192 MB at n = 6, P = 60.
static int fused_blocks_env() { const char* e = probe_env("RDYN_FUSED_BLOCKS"); return e ? atoi(e) : 256; }
static const int kFusedBlocks = 256;  // upper bound used for the workspace size; the launch uses fused_blocks_env()
extern "C" int rdyn_identification_gram(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b, const double* tau_meas,
                                        double* G, double* cvec, double* bb, int accumulate, void* workspace, size_t workspace_bytes);

// A chain whose own columns exceed what the Gram kernels hold (111) but whose REDUCED companion does not (fixed frames: 10 joints with
// 7 input joints = 100 columns + friction columns) is served through the companion like a chain longer than the kernels sweep.
static bool gram_only_through_reduced(const rdyn_chain* c, int n_comp_cols)
{
  return c->long_chain() || (c->reduced && rdyn_gram_blocks_for(10 * c->n_joints() + n_comp_cols) > 7);
}

// a chain with more input joints than the unrolled kernels sweep (no companion) whose own columns still fit the Gram kernel: 11 input
// joints (110 columns + tau_meas = 111) -- chunk images by rdyn_long_local.hip, contracted by k_gram
static bool gram_by_long_images(const rdyn_chain* c) { return c->long_chain() && !c->reduced && rdyn_gram_blocks_for(10 * c->n_joints()) <= 7; }

size_t rdyn_regressor_gram_workspace_bytes(const rdyn_chain* c, int64_t chunk_samples)
{
  if (!c) return 0;
  if (gram_by_long_images(c))
  {
    const int P = 10 * c->n_joints();
    return (gram_slab_bytes(P) + (size_t)default_chunk(chunk_samples) * c->n_active() * (P + 1) * sizeof(double) + 255) & ~(size_t)255;
  }
  if (gram_only_through_reduced(c, 0))
  {
    // only the reduced companion is swept: its workspace + its normal equations behind it
    const size_t w = c->reduced ? rdyn_regressor_gram_workspace_bytes(c->reduced.get(), 0) : 0;
    return w ? w + reduce_tmp_bytes(10 * c->reduced->n_joints()) : 0;
  }
  const int P = 10 * c->n_joints();
  if (rdyn_gram_blocks_for(P) > 7) return 0;
  const int64_t chunk = default_chunk(chunk_samples);
  const size_t chunked = (size_t)chunk * c->n_active() * (P + 1) * sizeof(double);
  const size_t fused = (size_t)kFusedBlocks * 256 * c->n_active() * (P + 1) * sizeof(double);  // one tile image per workgroup
  const size_t base = (gram_slab_bytes(P) + (chunked > fused ? chunked : fused) + 255) & ~(size_t)255;
  // chains with joints that are not input joints: the kernels run on the reduced companion, whose normal equations wait at the end
  return base + (c->reduced ? reduce_tmp_bytes(10 * c->reduced->n_joints()) : 0);
}

// Chains with non-input joints (default paths, chunk_samples <= 0): the regressor -> Gram kernels run on the reduced companion
// (rdyn_chain.hpp: every joint an input joint -- the fastest instantiations, 10 n instead of 10 nJ columns), then G = E' G_red E.
// chunk_samples > 0 keeps the chain as it is (the reference ordering of the two-kernel path).
static int gram_through_reduced(const rdyn_chain* c, const rdyn_component* comps, int n_comps, int K, const rdyn_batch* b, const double* tau_meas,
                                double* G, double* cvec, double* bb, int accumulate, void* workspace, size_t need_bytes)
{
  const rdyn_chain* r = c->reduced.get();
  const int Cr = 10 * r->n_joints() + K;
  const size_t tmp = reduce_tmp_bytes(Cr);
  double* Gr = (double*)((char*)workspace + need_bytes - tmp);
  int st = n_comps > 0 ? rdyn_identification_gram(r, comps, n_comps, b, tau_meas, Gr, Gr + (size_t)Cr * Cr, Gr + (size_t)Cr * Cr + Cr, 0, workspace, need_bytes - tmp)
                       : rdyn_regressor_gram(r, b, tau_meas, Gr, Gr + (size_t)Cr * Cr, Gr + (size_t)Cr * Cr + Cr, 0, 0, workspace, need_bytes - tmp);
  if (st != RDYN_OK) return st;
  RdynGramExpandArgs ea;
  memset(&ea, 0, sizeof ea);
  st = device_expand(c, &ea.X);
  if (st != RDYN_OK) return st;
  ea.G_red = Gr;
  ea.c_red = Gr + (size_t)Cr * Cr;
  ea.bb_red = Gr + (size_t)Cr * Cr + Cr;
  for (int f = 0; f < c->n_joints(); ++f) ea.red_of[f] = c->red_of[f];
  ea.n_joints = c->n_joints();
  ea.n_red = r->n_joints();
  ea.n_comp_cols = K;
  ea.add_to_output = accumulate ? 1 : 0;
  ea.G = G;
  ea.c = cvec;
  ea.bb = bb;
  RDYN_HIP_TRY(rdyn_launch_gram_expand(ea, (hipStream_t)b->stream));
  return RDYN_OK;
}

// the chain the tile kernels sweep: the sorted view when the input joints were listed out of chain order (rdyn_chain.hpp)
static const rdyn_chain* ordered(const rdyn_chain* c) { return c->sorted ? c->sorted.get() : c; }
// rows of the row waves of the one-lane-per-sample sweepers (rdyn_kin_sweepers.inc): row l (input joints in chain order) is carried
// through n - l links; longest row first, each to the wave with the least work so far.  The Gram kernel has seven row waves with two
// slots (the one that shares the kinematics wave's SIMD, `skip`, stays empty up to 6 joints), pass B of the R factor three with three.
static void fill_sw_rows(RdynLdsGramArgs* la, int n, int waves, int slots, int skip)
{
  for (int i = 0; i < 21; ++i) la->sw_rows[i] = 99;
  int load[7] = {0, 0, 0, 0, 0, 0, 0}, used[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int l = 0; l < n && l < 8; ++l)
  {
    int best = -1;
    for (int w = 0; w < waves; ++w)
      if (w != skip && used[w] < slots && (best < 0 || load[w] < load[best])) best = w;
    if (best < 0) return;
    la->sw_rows[3 * best + used[best]++] = l;
    load[best] += n - l;
  }
}
static void fill_in_map(const rdyn_chain* c, RdynLdsGramArgs* la)
{
  const int n = c->n_active();
  for (int r = 0; r < 8; ++r) la->in_map[r] = (r < n && r < (int)c->row_input.size()) ? c->row_input[r] : r;
  fill_sw_rows(la, n, 7, 2, !probe_env("RDYN_KIN_ROW4") && n <= 6 ? 3 : -1);
  la->sweep_lanes = 0;
}
// the one-lane-per-sample sweepers of the wave-pair Gram kernel serve this chain (every joint an input joint): the tile padding they
// use (4, or 2 = the compact layout), 0 = no
static int kin_sweeper_pad(const rdyn_chain* c, int n_comp_cols)
{
  if (probe_env("RDYN_GRAM_SWEEPER") && !strcmp(probe_env("RDYN_GRAM_SWEEPER"), "pair")) return 0;
  if (c->n_active() != c->n_joints()) return 0;
  return rdyn_regressor_gram_duo_kin_pad(c->n_joints(), n_comp_cols);
}

// Tile layout of the LDS-resident regressor -> Gram kernels (rdyn_lds_gram.hip, rdyn_pipe_gram.hip, rdyn_duo_gram.hip): the columns
// of link f keep the rows of the input joints at chain index <= f; K component columns (one 16-row group each) and the measured
// torque follow.  Returns false when the input joints are not in chain order (the packed rows must be a prefix).
static bool build_lds_tile(const rdyn_chain* c, int n_comp_cols, bool dummy_slot, RdynLdsGramArgs* la, bool compact = false)
{
  // column padding: 4 doubles keep the 16 lanes of an MFMA operand read on disjoint banks; the compact layout (2 doubles, 2-way
  // conflicts on the consumer's reads, which has slack) is what lets four 7-joint tiles WITH component columns fit 160 KB
  const int pad = compact ? 2 : 4;
  const int n = c->n_active(), nJ = c->n_joints();
  bool monotonic = true;
  for (int j = 1; j < n; ++j) monotonic = monotonic && c->active[j] > c->active[j - 1];
  la->all_revolute = probe_env("RDYN_DUO_NO_ALLREV") ? 0 : 1;
  for (int f = 0; f < nJ; ++f)
    if (c->host_joints[f].type != RDYN_REVOLUTE) la->all_revolute = 0;
  int off = 0;
  for (int f = 0; f < nJ; ++f)
  {
    int m = 0;
    for (int j = 0; j < n; ++j) m += (c->active[j] <= f) ? 1 : 0;
    la->lds_m[f] = m;
    la->lds_stride[f] = (16 * m + pad) * 8;
    la->lds_off[f] = off;
    off += 10 * la->lds_stride[f];
  }
  la->lds_off_c = off;
  la->comp_stride = (16 + pad) * 8;
  off += n_comp_cols * la->comp_stride;
  la->lds_off_b = off;
  off += (16 * n + pad) * 8;
  la->lds_dummy_off = off;
  if (dummy_slot) off += 64 * 8;
  la->tile_bytes = (off + 255) & ~255;
  la->n_active = n;
  for (int j = 0; j < n; ++j) la->first_col[j] = 10 * c->active[j];
  fill_in_map(c, la);
  return monotonic;
}

int rdyn_regressor_gram(const rdyn_chain* c, const rdyn_batch* b, const double* tau_meas, double* G, double* cvec, double* bb,
                        int accumulate, int64_t chunk_samples, void* workspace, size_t workspace_bytes)
{
  const bool long_images = c && gram_by_long_images(c);
  int st = check_batch(c, b, true, true, "rdyn_regressor_gram", long_images ? LONG_KERNELS : LONG_COMPANION);
  if (st != RDYN_OK) return st;
  if (!G || !workspace)
  {
    rdyn_set_error("rdyn_regressor_gram: null output or workspace");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const int n = c->n_active(), P = 10 * c->n_joints();
  if (long_images)
  {
    const int64_t chunk = default_chunk(chunk_samples), N = b->n_samples;
    if (workspace_bytes < rdyn_regressor_gram_workspace_bytes(c, chunk))
    {
      rdyn_set_error("rdyn_regressor_gram: workspace too small");
      return RDYN_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard g;
    st = g.enter(b->device);
    if (st != RDYN_OK) return st;
    hipStream_t stream = (hipStream_t)b->stream;
    if (N == 0)
    {
      if (!accumulate)
      {
        RDYN_HIP_TRY(hipMemsetAsync(G, 0, sizeof(double) * P * P, stream));
        if (cvec) RDYN_HIP_TRY(hipMemsetAsync(cvec, 0, sizeof(double) * P, stream));
        if (bb) RDYN_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(double), stream));
      }
      return RDYN_OK;
    }
    double* const slabs = (double*)workspace;
    double* const image = (double*)((char*)workspace + gram_slab_bytes(P));
    const int64_t in_step = (b->layout == RDYN_LAYOUT_SAMPLE_MAJOR) ? n : 1;
    for (int64_t s0 = 0; s0 < N; s0 += chunk)
    {
      const int64_t cnt = (N - s0 < chunk) ? (N - s0) : chunk;
      RdynLongLocalArgs a;
      memset(&a, 0, sizeof a);
      st = device_const_long(c, &a.chain_long);
      if (st != RDYN_OK) return st;
      a.q = b->q + s0 * in_step;
      a.dq = b->dq + s0 * in_step;
      a.ddq = b->ddq + s0 * in_step;
      a.bcol = tau_meas ? tau_meas + s0 * in_step : nullptr;
      a.bcol_col = P;
      a.n_samples = cnt;
      rec_strides(b, n, &a.in_ss, &a.in_sj);  // element-major: the joint stride stays the FULL batch's N
      a.Y = image;                             // dense element-major image of this chunk: rows j * cnt + s, lda = n * cnt
      a.y_ss = 1;
      a.y_sr = cnt;
      a.y_sc = (int64_t)n * cnt;
      RDYN_HIP_TRY(rdyn_launch_long_local(RDYN_MODE_REGRESSOR, c->n_joints(), a, stream));
      st = gram_launch(image, (int64_t)n * cnt, (int64_t)n * cnt, P, tau_meas ? image + (int64_t)P * n * cnt : nullptr, G, cvec, bb, s0 > 0 ? 1 : 0,
                       s0 + cnt >= N, accumulate ? 1 : 0, slabs, stream);
      if (st != RDYN_OK) return st;
    }
    return RDYN_OK;
  }
  if (gram_only_through_reduced(c, 0))
  {
    const size_t need = rdyn_regressor_gram_workspace_bytes(c, 0);
    if (need == 0 || workspace_bytes < need)
    {
      rdyn_set_error("rdyn_regressor_gram: %s", need == 0 ? "at most 111 columns of the reduced chain are supported" : "workspace too small");
      return need == 0 ? RDYN_ERR_UNSUPPORTED : RDYN_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard g;
    st = g.enter(b->device);
    if (st != RDYN_OK) return st;
    if (b->n_samples == 0)
    {
      if (!accumulate)
      {
        RDYN_HIP_TRY(hipMemsetAsync(G, 0, sizeof(double) * P * P, (hipStream_t)b->stream));
        if (cvec) RDYN_HIP_TRY(hipMemsetAsync(cvec, 0, sizeof(double) * P, (hipStream_t)b->stream));
        if (bb) RDYN_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(double), (hipStream_t)b->stream));
      }
      return RDYN_OK;
    }
    return gram_through_reduced(c, nullptr, 0, 0, b, tau_meas, G, cvec, bb, accumulate, workspace, need);
  }
  if (rdyn_gram_blocks_for(P) > 7)
  {
    rdyn_set_error("rdyn_regressor_gram: at most 111 regressor columns are supported");
    return RDYN_ERR_UNSUPPORTED;
  }
  const int64_t chunk = default_chunk(chunk_samples);
  if (workspace_bytes < rdyn_regressor_gram_workspace_bytes(c, chunk))
  {
    rdyn_set_error("rdyn_regressor_gram: workspace too small");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  const RdynChainConst* dc = nullptr;
  st = device_const(c, &dc);
  if (st != RDYN_OK) return st;
  hipStream_t stream = (hipStream_t)b->stream;
  double* slabs = (double*)workspace;
  double* scratch = (double*)((char*)workspace + gram_slab_bytes(P));
  const int64_t N = b->n_samples;
  const int64_t in_step = (b->layout == RDYN_LAYOUT_SAMPLE_MAJOR) ? n : 1;  // pointer advance per sample
  if (N == 0)
  {
    if (!accumulate)
    {
      RDYN_HIP_TRY(hipMemsetAsync(G, 0, sizeof(double) * P * P, stream));
      if (cvec) RDYN_HIP_TRY(hipMemsetAsync(cvec, 0, sizeof(double) * P, stream));
      if (bb) RDYN_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(double), stream));
    }
    return RDYN_OK;
  }
  if (c->reduced && chunk_samples <= 0 && !probe_env("RDYN_GRAM_NO_REDUCE"))
    return gram_through_reduced(c, nullptr, 0, 0, b, tau_meas, G, cvec, bb, accumulate, workspace, rdyn_regressor_gram_workspace_bytes(c, chunk));
  // structural zero band of every row block (input joint j): columns < 10 * chain index of joint j
  int first_col[RDYN_MAX_SWEPT_JOINTS];
  for (int j = 0; j < n; ++j) first_col[j] = 10 * c->active[j];
  const char* path_env = probe_env("RDYN_GRAM_PATH");  // A/B only: "lds" (default when eligible), "image", "two"
  const bool want_lds = !path_env || !strcmp(path_env, "lds") || !strcmp(path_env, "lds0") || !strcmp(path_env, "pipe") || !strcmp(path_env, "duo");
  if (chunk_samples <= 0 && !probe_env("RDYN_GRAM_UNFUSED") && want_lds && n >= 2 && n <= 8)
  {
    // LDS-resident path (rdyn_lds_gram.hip): needs input joints in chain order (rows of a link's columns are a prefix)
    // and four tiles inside 160 KB of LDS.
    RdynLdsGramArgs la;
    memset(&la, 0, sizeof la);
    // wave-pair kernel (rdyn_duo_gram.hip) by default; RDYN_GRAM_PATH=pipe / lds0 keep the single-wave kernels (A/B)
    const bool duo = rdyn_regressor_gram_duo_supported(P) && !(path_env && (!strcmp(path_env, "lds0") || !strcmp(path_env, "pipe")));
    const bool pipe = !duo && rdyn_regressor_gram_pipe_supported(P) && !(path_env && !strcmp(path_env, "lds0"));
    // input joints listed out of chain order: the sorted view is swept, every row's inputs read through its map
    const rdyn_chain* const co = ordered(c);
    const int kin_pad = duo ? kin_sweeper_pad(co, 0) : 0;
    const bool monotonic = build_lds_tile(co, 0, pipe, &la, kin_pad == 2);
    const int nb = rdyn_gram_blocks_for(P);
    size_t lds_bytes = 4 * (size_t)la.tile_bytes + (kin_pad ? RDYN_KIN_XCH_BYTES(co->n_joints()) : 0);
    la.sweep_lanes = kin_pad != 0;
    const size_t red_bytes = (size_t)(nb * (nb + 1) / 2) * 256 * sizeof(double);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    if (monotonic && lds_bytes <= 160 * 1024)
    {
      st = device_const(co, &la.chain);
      if (st != RDYN_OK) return st;
      la.q = b->q;
      la.dq = b->dq;
      la.ddq = b->ddq;
      la.bcol = tau_meas;
      la.n_samples = N;
      rec_strides(b, n, &la.in_ss, &la.in_sj);
      la.slabs = slabs;
      if (const char* dbg = probe_env("RDYN_FUSED_DEBUG")) la.debug = atoi(dbg);
      const int64_t tiles = (N + 15) / 16;
      int want = fused_blocks_env();
      if (want < 1 || want > kFusedBlocks) want = kFusedBlocks;
      const int blocks = (int)((tiles + 3) / 4 < want ? (tiles + 3) / 4 : want);
      if (duo)
        RDYN_HIP_TRY(rdyn_launch_regressor_gram_duo(P, la, blocks, lds_bytes, stream));
      else if (pipe)
        RDYN_HIP_TRY(rdyn_launch_regressor_gram_pipe(P, la, blocks, lds_bytes, stream));
      else
        RDYN_HIP_TRY(rdyn_launch_regressor_gram_lds(P, la, blocks, lds_bytes, stream));
      RdynGramArgs ga;
      memset(&ga, 0, sizeof ga);
      ga.P = P;
      ga.add_to_output = accumulate ? 1 : 0;
      ga.slabs = slabs;
      ga.G = G;
      ga.c = cvec;
      ga.bb = bb;
      ga.desc_nj = duo ? c->n_joints() : 0;  // the wave-pair kernel accumulates in descending link order
      RDYN_HIP_TRY(rdyn_launch_gram_finish(ga, blocks, stream));
      return RDYN_OK;
    }
  }
  if (chunk_samples <= 0 && !probe_env("RDYN_GRAM_UNFUSED") && (!path_env || strcmp(path_env, "two")))
  {
    // default: ONE persistent kernel, the regressor image never goes through HBM (rdyn_fused_gram.hip).
    // chunk_samples > 0 selects the two-kernel chunked path below (kept for A/B and as the reference ordering).
    RdynFusedGramArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.sweep.chain = dc;
    fa.sweep.q = b->q;
    fa.sweep.dq = b->dq;
    fa.sweep.ddq = b->ddq;
    fa.sweep.bcol = tau_meas;
    fa.sweep.n_samples = N;
    rec_strides(b, n, &fa.sweep.in_ss, &fa.sweep.in_sj);
    fa.n_active = n;
    for (int j = 0; j < n; ++j) fa.first_col[j] = first_col[j];
    fa.images = scratch;
    fa.slabs = slabs;
    if (const char* dbg = probe_env("RDYN_FUSED_DEBUG")) fa.debug = atoi(dbg);
    const int64_t tiles = (N + 255) / 256;
    int want = fused_blocks_env();
    if (want < 1 || want > kFusedBlocks) want = kFusedBlocks;
    const int blocks = (int)(tiles < want ? tiles : want);
    // structural zeros that the sweep never stores but the Gram still loads must read as zero
    RDYN_HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)blocks * 256 * n * (P + 1), stream));
    RDYN_HIP_TRY(rdyn_launch_regressor_gram_fused(c->n_joints(), fa, blocks, stream));
    RdynGramArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.P = P;
    ga.add_to_output = accumulate ? 1 : 0;
    ga.slabs = slabs;
    ga.G = G;
    ga.c = cvec;  // without tau_meas the image's column P stays zero: c = 0, bb = 0
    ga.bb = bb;
    RDYN_HIP_TRY(rdyn_launch_gram_finish(ga, blocks, stream));
    return RDYN_OK;
  }
  int64_t prev_cnt = -1;
  for (int64_t s0 = 0; s0 < N; s0 += chunk)
  {
    const int64_t cnt = (N - s0 < chunk) ? (N - s0) : chunk;
    if (cnt != prev_cnt)
    {
      // The sweep kernel does not store the zeros that k_gram never loads; where the image layout changes (first
      // chunk, shorter last chunk) positions that are unwritten zeros must not hold stale data (16-row groups that
      // straddle two row blocks read a few of them).
      RDYN_HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)cnt * n * (P + 1), stream));
      prev_cnt = cnt;
    }
    RdynSweepArgs a;
    memset(&a, 0, sizeof a);
    a.chain = dc;
    a.q = b->q + s0 * in_step;
    a.dq = b->dq + s0 * in_step;
    a.ddq = b->ddq + s0 * in_step;
    a.bcol = tau_meas ? tau_meas + s0 * in_step : nullptr;
    a.n_samples = cnt;
    rec_strides(b, n, &a.in_ss, &a.in_sj);  // element-major: joint stride stays the FULL batch's N
    a.Y = scratch;                          // element-major image of this chunk: rows j * cnt + s, lda = n * cnt
    a.y_ss = 1;
    a.y_sr = cnt;
    a.y_sc = (int64_t)n * cnt;
    RDYN_HIP_TRY(rdyn_launch_local_sweep(c->n_joints(), RDYN_MODE_REGRESSOR_GRAM, a, stream));
    const bool last = (s0 + cnt >= N);
    st = gram_launch(scratch, (int64_t)n * cnt, (int64_t)n * cnt, P, tau_meas ? scratch + (int64_t)P * n * cnt : nullptr, G, cvec, bb,
                     s0 > 0 ? 1 : 0, last, accumulate ? 1 : 0, slabs, stream, cnt, first_col, n);
    if (st != RDYN_OK) return st;
  }
  return RDYN_OK;
}

// ---- tall-skinny QR: the R factor without forming A'A (rdyn_tsqr.hip, rdyn_tsqr_wide.hip, rdyn_cholqr.hip) ---------------------
static const int kTsqrBlocks = 256;    // persistent workgroups (one per CU), four waves = four running factors each
static const int kCholqrBlocks = 256;
// ~0.2 ms of fixed cost (the subsample's Gram matrix, the two small dense kernels, seven launches that leave at once) + 0.78 / 1.2 ms
// per 1e6 samples (6 / 7 joints) against 0.3 - 0.4 ms + 2.8 / 3.8 ms per 1e6 samples for the Householder route: faster at every
// size measured (2 000 samples: 183 vs 288 us; 200 000: 364 vs 969 us).  Small batches keep the Householder folds -- a few hundred
// rows say little about what a preconditioner built on them is worth, and there is nothing to win.
static const int64_t kCholqrMinTiles = 256;        // fused routes: 4 096 samples
static const int64_t kCholqrMinGroups = 2048;      // rdyn_tsqr: 32 768 rows
static const int64_t kTsqrImageChunk = 65536;      // samples per chunk image of the 9 .. 10-joint factor route (0.53 GB at 10 joints; 16 384: 3x the time per sample -- every chunk pays the fixed cost of a factor call)

// offsets (doubles) of the regions every factor call carves out of its workspace.  Region 1: the Householder route's leaves + tree
// levels (rdyn_tsqr.hip or rdyn_tsqr_wide.hip).  Region 2: the preconditioned route's slabs, W, the intermediate factors and the flags.
struct TsqrLayout
{
  size_t householder_doubles = 0;
  size_t slabs = 0, w = 0, v = 0, r1p = 0, g2 = 0, r_swept = 0, r_full = 0, flag = 0, wide = 0, wide_doubles = 0, total_doubles = 0;
};
// n1: width of the factor that is computed (padded width where the kernels pad); nb: 16-column blocks of the preconditioned route's
// column space (0: that route does not serve the shape); n1_full: width of the expanded factor (0: nothing is expanded)
static TsqrLayout tsqr_layout(size_t householder_doubles, int n1, int nb, int n1_full)
{
  TsqrLayout L;
  L.householder_doubles = householder_doubles;
  size_t off = (L.householder_doubles + 31) & ~(size_t)31;
  auto take = [&](size_t doubles) {
    const size_t at = off;
    off = (off + doubles + 31) & ~(size_t)31;
    return at;
  };
  const int nt = nb * (nb + 1) / 2;
  L.slabs = take((size_t)kCholqrBlocks * nt * 256);
  L.w = take((size_t)nt * 256);
  L.r1p = take((size_t)n1 * n1);
  L.g2 = take((size_t)n1 * n1 + 1);
  L.r_swept = take((size_t)n1 * n1);
  L.v = take((size_t)n1 * n1);
  L.r_full = take((size_t)n1_full * n1_full);  // the expanded factor of a call that accumulates (folded into the caller's afterwards)
  L.flag = take(96);  // ints: [0] run round 1, [1] run the stand-by (Householder) call, [2] run round 0, [16 .. 127] the deferred columns;
                      // doubles [64 .. 69]: gamma (preconditioner), rho, gamma (factor kernel) of the two rounds (diagnostics) -- behind
                      // the 112 column flags (until round 5 at [56 .. 61], on top of the flags of columns 96 .. 107)
  L.wide_doubles = (nb > 0 && n1 > rdyn_cholqr_max_cols_lds()) ? (size_t)2 * n1 * n1 : 0;
  L.wide = take(L.wide_doubles);  // the dense kernels' two n1 x n1 squares where they do not fit the LDS (97 .. 112 columns)
  L.total_doubles = off;
  return L;
}

// The preconditioned CholeskyQR rounds on a Gram matrix of a row subsample that already sits in ws + L.g2 ([G (P x P) | c (P) | bb]):
// two rounds of  precond -> pass B (run_pass_b(W, run_flag, slabs)) -> slab reduction -> factor kernel, the second one and the
// stand-by started by the device only (flags).  R_out <- the accepted factor (n1 x n1).  Shared by the fused routes and rdyn_tsqr.
typedef std::function<int(const double* W, const int* run_flag, double* slabs)> PassB;
static int cholqr_rounds(double* ws, const TsqrLayout& L, int n1, int col_shift, int nb, int slab_nb, int has_b, double row_scale, int blocks,
                         double* R_out, hipStream_t stream, const PassB& run_pass_b)
{
  int* const flag = (int*)(ws + L.flag);
  RdynGramArgs ga;
  memset(&ga, 0, sizeof ga);
  ga.P = n1 - 1;
  ga.slabs = ws + L.slabs;
  ga.G = ws + L.g2;
  ga.c = ws + L.g2 + (size_t)(n1 - 1) * (n1 - 1);
  ga.bb = ga.c + (n1 - 1);
  ga.col_shift = col_shift;
  ga.slab_nb = slab_nb;
  const int n_rounds = probe_env("RDYN_CHOLQR_ROUNDS") ? atoi(probe_env("RDYN_CHOLQR_ROUNDS")) : 2;  // A/B builds only
  double* const wide_sq = L.wide_doubles ? ws + L.wide : nullptr;  // factors beyond 96 columns: the dense kernels' two squares
  for (int round = 0; round < n_rounds; ++round)
  {
    // round 0: W from the subsample's Gram matrix.  Round 1 (CholeskyQR2 on top): W from round 0's factor; its kernels leave at once
    // unless round 0's factor kernel asked for it (flag[0]).  Round 0 itself runs when its preconditioner is fit for it (flag[2]).
    const int* const run = round == 0 ? flag + 2 : flag;
    if (round == 0)
      RDYN_HIP_TRY(rdyn_launch_cholqr_precond(nullptr, ga.G, ga.c, ga.bb, n1, col_shift, nb, row_scale, ws + L.r1p, ws + L.w, ws + L.v, flag + 16, flag, 0,
                                              nullptr, ws + L.flag + 64, stream, wide_sq));
    else
      RDYN_HIP_TRY(rdyn_launch_cholqr_precond(R_out, nullptr, nullptr, nullptr, n1, col_shift, nb, 1.0, ws + L.r1p, ws + L.w, ws + L.v, flag + 16, flag, 1,
                                              flag, ws + L.flag + 65, stream, wide_sq));
    int st = run_pass_b(ws + L.w, run, ws + L.slabs);
    if (st != RDYN_OK) return st;
    ga.run_flag = run;
    RDYN_HIP_TRY(rdyn_launch_gram_finish(ga, blocks, stream));
    RDYN_HIP_TRY(rdyn_launch_cholqr_factor(ga.G, ga.c, ga.bb, n1, has_b, ws + L.r1p, ws + L.v, flag + 16, R_out, flag, round, run, ws + L.flag + 66 + round,
                                           stream, wide_sq));
  }
  return RDYN_OK;
}

static void read_report(const double* raw, int n1, rdyn_tsqr_report* out)
{
  int flags[128];
  memcpy(flags, raw, sizeof flags);
  out->route = 1;
  const bool round0 = flags[2] != 0, round1 = round0 && flags[0] != 0;
  out->stage = flags[1] ? 2 : (round1 ? 1 : 0);
  for (int k = 0; k < n1 && 16 + k < 128; ++k) out->n_deferred += flags[16 + k] ? 1 : 0;
  if (round0)
  {
    out->gamma[0] = raw[68];
    out->rho[0] = raw[66];
  }
  if (round1)
  {
    out->gamma[1] = raw[69];
    out->rho[1] = raw[67];
  }
}

// ---- rdyn_tsqr: a materialised matrix
struct TsqrRowsPlan
{
  int n1 = 0, nc_reg = 0, nb = 0;  // nc_reg: padded width of the register-resident folds (0: wider than 64 -> LDS-resident folds)
  bool cholqr_ok = false;          // the dense steps of the preconditioned route hold the factor
  TsqrLayout L;
};
static bool tsqr_rows_plan(int n1, TsqrRowsPlan* p)
{
  if (n1 < 1 || n1 > rdyn_tsqr_wide_max_cols()) return false;
  p->n1 = n1;
  p->nc_reg = rdyn_tsqr_padded_cols(n1);
  p->cholqr_ok = n1 >= 2 && n1 <= rdyn_cholqr_max_cols();
  p->nb = p->cholqr_ok ? (n1 + 15) / 16 : 0;
  size_t hh = rdyn_tsqr_wide_workspace_doubles(n1, kTsqrBlocks);
  if (p->nc_reg)
  {
    const size_t reg = rdyn_tsqr_workspace_doubles(p->nc_reg, kTsqrBlocks);
    if (reg > hh) hh = reg;
  }
  p->L = tsqr_layout(hh, n1, p->nb, 0);
  return true;
}

size_t rdyn_tsqr_workspace_bytes(int n_cols_with_rhs)
{
  TsqrRowsPlan p;
  return tsqr_rows_plan(n_cols_with_rhs, &p) ? p.L.total_doubles * sizeof(double) : 0;
}

int rdyn_tsqr(const double* A, int64_t rows, int64_t lda, int n_cols, const double* bvec, double* R, int accumulate, void* workspace,
              size_t workspace_bytes, int device, void* stream_v)
{
  if (!A || !R || rows < 0 || lda < rows || n_cols < 1 || !workspace)
  {
    rdyn_set_error("rdyn_tsqr: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const int n1 = n_cols + (bvec ? 1 : 0);
  TsqrRowsPlan p;
  if (!tsqr_rows_plan(n1, &p))
  {
    rdyn_set_error("rdyn_tsqr: at most %d columns (right-hand side included) are supported", rdyn_tsqr_wide_max_cols());
    return RDYN_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < p.L.total_doubles * sizeof(double))
  {
    rdyn_set_error("rdyn_tsqr: workspace too small");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g;
  int st = g.enter(device);
  if (st != RDYN_OK) return st;
  hipStream_t stream = (hipStream_t)stream_v;
  if (rows == 0)
  {
    if (!accumulate) RDYN_HIP_TRY(hipMemsetAsync(R, 0, sizeof(double) * n1 * n1, stream));
    return RDYN_OK;
  }
  double* const ws = (double*)workspace;
  const TsqrLayout& L = p.L;
  // the Householder folds of all rows: in registers up to 64 columns, with the factor in LDS beyond
  auto householder = [&](double* R_to, int acc, const int* run_flag, int fan) -> int {
    if (p.nc_reg)
    {
      const int64_t blocks64 = (rows + 63) / 64;  // one wave folds 64 rows at a time
      const int blocks = (int)((blocks64 + 3) / 4 < kTsqrBlocks ? (blocks64 + 3) / 4 : kTsqrBlocks);
      RDYN_HIP_TRY(rdyn_launch_tsqr_rows(A, bvec, rows, lda, n_cols, blocks, ws, R_to, acc, stream, run_flag, fan));
    }
    else
      RDYN_HIP_TRY(rdyn_launch_tsqr_wide_rows(A, bvec, rows, lda, n_cols, kTsqrBlocks, ws, R_to, acc, run_flag, stream));
    return RDYN_OK;
  };
  const int64_t groups = (rows + 15) / 16;
  const char* route_env = probe_env("RDYN_TSQR_ROUTE");  // A/B builds only: "householder" / "cholqr"
  bool cholqr = p.cholqr_ok && groups >= kCholqrMinGroups;
  if (route_env && p.cholqr_ok) cholqr = !strcmp(route_env, "cholqr");
  if (!cholqr) return householder(R, accumulate ? 1 : 0, nullptr, 2);
  // ---- preconditioned CholeskyQR on the matrix cores (rdyn_cholqr.hip), as the fused routes below: the Gram matrix of every S-th
  // 16-row group (about 8 192 groups), the rounds, the stand-by
  const int64_t kSubGroups = probe_env("RDYN_CHOLQR_SUBTILES") ? atoll(probe_env("RDYN_CHOLQR_SUBTILES")) * 8 : 8192;
  int64_t gs = groups / kSubGroups > 1 ? groups / kSubGroups : 1;
  if (gs > 1 && gs % 2 == 0) ++gs;  // odd: does not lock onto power-of-two periods of the rows
  const int64_t sub_groups = (groups + gs - 1) / gs;
  int* const flag = (int*)(ws + L.flag);
  {
    RdynGramArgs sa;
    memset(&sa, 0, sizeof sa);
    // [A | b] as ONE matrix of n1 columns whose last column is split off as the right-hand side of the dense steps
    sa.A = A;
    sa.b = bvec ? bvec : A + (int64_t)(n1 - 1) * lda;
    sa.rows = rows;
    sa.lda = lda;
    sa.P = n1 - 1;
    sa.slabs = ws + L.slabs;
    sa.G = ws + L.g2;
    sa.c = ws + L.g2 + (size_t)(n1 - 1) * (n1 - 1);
    sa.bb = sa.c + (n1 - 1);
    sa.group_stride = (int)gs;
    const int sub4 = (int)((sub_groups + 3) / 4 < kCholqrBlocks ? (sub_groups + 3) / 4 : kCholqrBlocks);
    RDYN_HIP_TRY(rdyn_launch_gram(sa, sub4, stream));
    RDYN_HIP_TRY(rdyn_launch_gram_finish(sa, sub4, stream));
  }
  double* const R_new = accumulate ? ws + L.r_swept : R;
  const int blocks = (int)((groups + 3) / 4 < kCholqrBlocks ? (groups + 3) / 4 : kCholqrBlocks);
  st = cholqr_rounds(ws, L, n1, 0, p.nb, p.nb, 1, sqrt((double)groups / (double)sub_groups), blocks, R_new, stream,
                     [&](const double* W, const int* run, double* slabs) -> int {
                       RDYN_HIP_TRY(rdyn_launch_pgram_rows(A, bvec, rows, lda, n_cols, W, slabs, run, blocks, stream));
                       return RDYN_OK;
                     });
  if (st != RDYN_OK) return st;
  st = householder(R_new, 0, flag + 1, 16);  // stand-by: started by the device only when neither round was accepted
  if (st != RDYN_OK) return st;
  if (accumulate) RDYN_HIP_TRY(rdyn_launch_cholqr_fold(R_new, R, n1, stream));
  return RDYN_OK;
}

int rdyn_tsqr_rows_last_report(int n_cols_with_rhs, int64_t rows, const void* workspace, int device, void* stream, rdyn_tsqr_report* out)
{
  TsqrRowsPlan p;
  if (!workspace || !out || rows < 0 || !tsqr_rows_plan(n_cols_with_rhs, &p))
  {
    rdyn_set_error("rdyn_tsqr_rows_last_report: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  memset(out, 0, sizeof *out);
  if (!p.cholqr_ok || (rows + 15) / 16 < kCholqrMinGroups) return RDYN_OK;  // route 0: the Householder folds, nothing to report
  DeviceGuard g;
  int st = g.enter(device);
  if (st != RDYN_OK) return st;
  double raw[96];
  RDYN_HIP_TRY(hipMemcpyAsync(raw, (const double*)workspace + p.L.flag, sizeof raw, hipMemcpyDeviceToHost, (hipStream_t)stream));
  RDYN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  read_report(raw, p.n1, out);
  return RDYN_OK;
}

// ---- rdyn_regressor_tsqr / rdyn_identification_tsqr: rows generated by the sweep, never stored
// the chain whose rows are swept: the reduced companion when the chain has non-input joints (component columns belong to input
// joints, which the companion keeps in the same order: they ride along unchanged).  A companion of ONE joint is below what the
// sweeping kernels are built for: such a chain is swept as it is.

// rectangular 16-sample tile of rdyn_tsqr_wide.hip: every column 16 n rows + 4 doubles, structural zeros stored
static bool build_rect_tile(const rdyn_chain* c, int n_comp_cols, RdynLdsGramArgs* la)
{
  const int n = c->n_active(), nJ = c->n_joints();
  const int cs = (16 * n + 4) * 8;
  bool monotonic = true;
  for (int j = 1; j < n; ++j) monotonic = monotonic && c->active[j] > c->active[j - 1];
  la->all_revolute = 0;
  for (int f = 0; f < nJ; ++f)
  {
    int m = 0;
    for (int j = 0; j < n; ++j) m += (c->active[j] <= f) ? 1 : 0;
    la->lds_m[f] = m;
    la->lds_stride[f] = cs;
    la->lds_off[f] = 10 * f * cs;
  }
  la->lds_off_c = 10 * nJ * cs;
  la->comp_stride = cs;
  la->comp_row_step = 128;
  la->lds_off_b = (10 * nJ + n_comp_cols) * cs;
  la->lds_dummy_off = 0;
  la->tile_bytes = (10 * nJ + n_comp_cols + 1) * cs;
  la->n_active = n;
  for (int j = 0; j < n; ++j) la->first_col[j] = 10 * c->active[j];
  fill_in_map(c, la);
  return monotonic;
}

// everything a factor call decides from (chain, components) alone: which chain is swept, which kernels serve it, the workspace
struct TsqrPlan
{
  const rdyn_chain* cs = nullptr;
  bool expand = false;
  int n = 0, nJ = 0, K = 0, xb = 0;
  int n1s = 0, n1 = 0;        // factor widths: swept chain, chain
  int nc_reg = 0;             // padded factor width of the register-resident folds of rdyn_tsqr.hip (0: they do not serve the shape)
  bool wide = false;          // the LDS-resident folds of rdyn_tsqr_wide.hip serve the shape
  int pairs = 0, nb = 0;      // preconditioned route: pass-B configuration (0: not served), 16-column blocks of its column space
  bool sub_compact = false;   // the subsample pass (four tiles per workgroup) needs the compact tile
  // 9 .. 10 input joints (more rows per sample than the tile kernels' sweepers hold): the rows go through a chunk image in the
  // workspace -- swept by the one-thread-per-sample kernel, factored by rdyn_tsqr's kernels, chunk after chunk
  bool image = false;
  bool long_image = false;    // ... of a chain with 11 input joints (no companion): the chunk image comes from rdyn_long_local.hip
  size_t img_off = 0, rows_off = 0;  // doubles: the chunk image, rdyn_tsqr's own workspace
  RdynLdsGramArgs la, la_sub, la_wide;
  RdynLdsGramArgs la_b;       // pass B's tile: la, or the layout of the one-lane-per-sample sweepers (sweep_lanes set)
  TsqrLayout L;
  const char* why = nullptr;  // when nothing serves the shape
};

static bool tsqr_plan(const rdyn_chain* c, const rdyn_component* comps, int n_comps, TsqrPlan* p)
{
  // the chain whose rows are swept: the reduced companion where the chain has joints that are not input joints; the tile kernels sweep
  // its sorted view (input joints listed out of chain order)
  const rdyn_chain* const raw = (c->reduced && c->reduced->n_joints() >= 2) ? c->reduced.get() : c;
  p->cs = ordered(raw);
  p->expand = raw != c;
  if (c->long_chain() && !p->expand)
  {
    // no companion (more than 10 input joints): served as long as the factor fits the widest one the kernels hold -- 11 input joints
    // without component columns (110 + 1 columns) -- through chunk images of the run-time-length regressor kernel
    const int K0 = n_comps > 0 ? rdyn_components_columns(comps, n_comps) : 0;
    if (K0 != 0 || 10 * c->n_joints() + 1 > rdyn_tsqr_wide_max_cols() || c->n_active() != c->n_joints())
    {
      p->why = "a chain of more than 10 joints needs 2 .. 10 input joints (11 without fixed joints and component columns)";
      return false;
    }
    p->cs = c;
    p->n = c->n_active();
    p->nJ = c->n_joints();
    p->K = 0;
    p->xb = 0;
    p->n1s = p->n1 = 10 * p->nJ + 1;
    memset(&p->la, 0, sizeof p->la);
    memset(&p->la_sub, 0, sizeof p->la_sub);
    memset(&p->la_wide, 0, sizeof p->la_wide);
    memset(&p->la_b, 0, sizeof p->la_b);
    p->image = p->long_image = true;
    p->L = tsqr_layout(0, p->n1s, 0, 0);
    p->img_off = (p->L.total_doubles + 31) & ~(size_t)31;
    p->rows_off = p->img_off + (size_t)kTsqrImageChunk * p->n * p->n1s;
    p->L.total_doubles = p->rows_off + rdyn_tsqr_workspace_bytes(p->n1s) / sizeof(double);
    return true;
  }
  const rdyn_chain* cs = p->cs;
  p->n = cs->n_active();
  p->nJ = cs->n_joints();
  const int K = n_comps > 0 ? rdyn_components_columns(comps, n_comps) : 0;
  if (K < 0 || K > 96)
  {
    p->why = "invalid components";
    return false;
  }
  p->K = K;
  p->xb = K > 0 ? 1 : 0;
  p->n1s = 10 * p->nJ + K + 1;
  p->n1 = 10 * c->n_joints() + K + 1;
  memset(&p->la, 0, sizeof p->la);
  memset(&p->la_sub, 0, sizeof p->la_sub);
  memset(&p->la_wide, 0, sizeof p->la_wide);
  memset(&p->la_b, 0, sizeof p->la_b);
  if (p->n > 8 && p->n <= RDYN_MAX_SWEPT_JOINTS && p->n == p->nJ && p->n1s <= rdyn_tsqr_wide_max_cols() &&
      !(p->expand && rdyn_cholqr_expand_lds_bytes(c->n_joints(), p->nJ, K) > 156 * 1024))
  {
    p->image = true;
    p->cs = raw;  // (the image's rows are the caller's input indices: the one-thread-per-sample sweep and the component kernel agree on that)
    p->L = tsqr_layout(0, p->n1s, 0, p->expand ? p->n1 : 0);
    p->img_off = (p->L.total_doubles + 31) & ~(size_t)31;
    p->rows_off = p->img_off + (size_t)kTsqrImageChunk * p->n * p->n1s;
    p->L.total_doubles = p->rows_off + rdyn_tsqr_workspace_bytes(p->n1s) / sizeof(double);
    return true;
  }
  if (p->n < 1 || p->n > 8 || p->nJ < 1 || !build_rect_tile(cs, K, &p->la_wide))
  {
    p->why = "chains of 1..10 input joints whose factor (after the reduction) has at most 112 columns are supported";
    return false;
  }
  const bool tile_ok = build_lds_tile(cs, K, false, &p->la) && 4 * (size_t)p->la.tile_bytes <= 160 * 1024;
  // register-resident Householder folds: 2..7 joints, 2..6 with component columns (which must fit one 16-column slot)
  p->nc_reg = (tile_ok && p->nJ >= 2 && p->nJ <= 7) ? rdyn_regressor_tsqr_cols(p->nJ, K) : 0;
  // LDS-resident folds: the rectangular tile beside the packed factor
  p->wide = rdyn_regressor_tsqr_wide_lds_bytes(p->n1s, p->n) != 0;
  // preconditioned route: every joint of the swept chain an input joint, the factor within the dense kernels' LDS, the component
  // columns within the one extra column block, and a subsample kernel for the shape
  p->nb = (10 * p->nJ + 1 + 15) / 16 + p->xb;
  if (p->n == p->nJ && p->nJ >= 2 && p->nJ <= 7 && p->n1s <= rdyn_cholqr_max_cols_lds() && p->n1s <= 16 * p->nb &&
      (K == 0 || rdyn_regressor_gram_duo_supports_components(10 * p->nJ, K)) && build_lds_tile(cs, K, false, &p->la))
  {
    p->pairs = rdyn_cholqr_pairs(p->nJ, p->la.tile_bytes, p->xb);
    if (p->pairs == -1)
    {
      // every wave sweeps and consumes its own COMPACT tile (k_regressor_pgram_solo): four of them fill the LDS, W stays in global memory
      RdynLdsGramArgs compact;
      memset(&compact, 0, sizeof compact);
      build_lds_tile(cs, K, false, &compact, true);
      if (4 * (size_t)compact.tile_bytes <= 160 * 1024)
        p->la = compact;
      else
        p->pairs = rdyn_cholqr_pairs(p->nJ, -p->la.tile_bytes, p->xb);  // too many component columns: two pairs beside W, or nothing
    }
    p->la_b = p->la;
    const int kin_pad = rdyn_cholqr_kin_pad(p->nJ, p->xb, p->pairs);
    if (kin_pad && !(probe_env("RDYN_PGRAM_SWEEPER") && !strcmp(probe_env("RDYN_PGRAM_SWEEPER"), "pair")))
    {
      // one lane per sample: wave 0 the link kinematics, three row waves (7 joints: four compact tiles + the exchange area, W in global memory)
      RdynLdsGramArgs kin;
      memset(&kin, 0, sizeof kin);
      build_lds_tile(cs, K, false, &kin, kin_pad == 2);
      const size_t need = (p->pairs == 4 ? rdyn_cholqr_w_doubles(p->nJ, p->xb) * 8 : 0) + 4 * (size_t)kin.tile_bytes + RDYN_KIN_XCH_BYTES_XV(p->nJ <= 6 ? 21 : 12);
      if (need <= 160 * 1024)
      {
        fill_sw_rows(&kin, p->n, 3, 3, -1);
        kin.sweep_lanes = 1;
        p->la_b = kin;
      }
    }
    if (p->pairs != 0)
    {
      build_lds_tile(cs, K, false, &p->la_sub);
      if (4 * (size_t)p->la_sub.tile_bytes > 160 * 1024)
      {
        build_lds_tile(cs, K, false, &p->la_sub, true);
        p->sub_compact = true;
        if (4 * (size_t)p->la_sub.tile_bytes > 160 * 1024) p->pairs = 0;
      }
    }
  }
  if (!p->nc_reg && !p->wide)
  {
    p->why = "the factor does not fit the LDS-resident folds (at most 112 columns)";
    return false;
  }
  if (p->expand && rdyn_cholqr_expand_lds_bytes(c->n_joints(), p->nJ, K) > 156 * 1024)
  {
    p->why = "the expansion of the reduced chain's factor exceeds the LDS of one workgroup";
    return false;
  }
  size_t hh = p->wide ? rdyn_tsqr_wide_workspace_doubles(p->n1s, kTsqrBlocks) : 0;
  int n1w = p->n1s;
  if (p->nc_reg)
  {
    const size_t reg = rdyn_tsqr_workspace_doubles(p->nc_reg, kTsqrBlocks);
    if (reg > hh) hh = reg;
    if (p->nc_reg > n1w) n1w = p->nc_reg;
  }
  p->L = tsqr_layout(hh, n1w, p->pairs != 0 ? p->nb : 0, p->expand ? p->n1 : 0);
  return true;
}

// swept_only: stop at the factor of the SWEPT chain (n1s x n1s into R, no expansion, no accumulation): the multi-device form gathers
// and folds these -- the smaller payload, and no limit on the width of the expanded factor -- and expands once per device
static int regressor_tsqr_run(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b, const double* tau_meas, double* R,
                              int accumulate, void* workspace, size_t workspace_bytes, const char* who, bool swept_only = false)
{
  // (a long chain without a companion: tsqr_plan decides -- 11 input joints are served through chunk images)
  int st = check_batch(c, b, true, true, who, (c && c->long_chain() && !c->reduced) ? LONG_KERNELS : LONG_COMPANION);
  if (st != RDYN_OK) return st;
  if (!R || !workspace || n_comps < 0 || n_comps > RDYN_MAX_COMPONENTS || (n_comps > 0 && !comps))
  {
    rdyn_set_error("%s: null output / workspace, or more than %d components", who, RDYN_MAX_COMPONENTS);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  TsqrPlan p;
  if (!tsqr_plan(c, comps, n_comps, &p))
  {
    rdyn_set_error("%s: %s", who, p.why);
    return RDYN_ERR_UNSUPPORTED;
  }
  const rdyn_chain* cs = p.cs;  // chains with fixed joints: the reduced companion is swept, the factor expanded
  const bool expand = p.expand && !swept_only;
  if (swept_only) accumulate = 0;
  const int n = p.n, nJ = p.nJ, K = p.K, n1s = p.n1s, n1 = p.n1;
  const TsqrLayout& L = p.L;
  if (workspace_bytes < L.total_doubles * sizeof(double))
  {
    rdyn_set_error("%s: workspace too small", who);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  RdynComponentArgs ca;
  memset(&ca, 0, sizeof ca);
  if (n_comps > 0)
  {
    st = fill_components(comps, n_comps, n, &ca);
    if (st != RDYN_OK) return st;
  }
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  const RdynChainConst* dc = nullptr;
  const RdynLongChainConst* dcl = nullptr;
  st = p.long_image ? device_const_long(cs, &dcl) : device_const(cs, &dc);
  if (st != RDYN_OK) return st;
  hipStream_t stream = (hipStream_t)b->stream;
  if (b->n_samples == 0)
  {
    if (!accumulate) RDYN_HIP_TRY(hipMemsetAsync(R, 0, sizeof(double) * (swept_only ? (size_t)n1s * n1s : (size_t)n1 * n1), stream));
    return RDYN_OK;
  }
  if (p.image)
  {
    // ---- 9 .. 10 input joints: [Y | C | tau_meas] of a chunk as an element-major image (rows j * cnt + s), factored by rdyn_tsqr's
    // kernels (matrix cores up to 96 columns, LDS-resident Householder folds beyond) and folded into the running factor
    double* const ws = (double*)workspace;
    double* const img = ws + p.img_off;
    double* const R_swept = expand ? ws + L.r_swept : R;
    const int cols = n1s - 1;  // columns in front of the right-hand side
    const int64_t in_step = (b->layout == RDYN_LAYOUT_SAMPLE_MAJOR) ? n : 1;
    int64_t prev_cnt = -1;
    for (int64_t s0 = 0; s0 < b->n_samples; s0 += kTsqrImageChunk)
    {
      const int64_t cnt = (b->n_samples - s0 < kTsqrImageChunk) ? (b->n_samples - s0) : kTsqrImageChunk;
      if (cnt != prev_cnt)
      {
        // the sweep does not store the structural zeros of the image: they must read as zeros
        RDYN_HIP_TRY(hipMemsetAsync(img, 0, sizeof(double) * (size_t)cnt * n * n1s, stream));
        prev_cnt = cnt;
      }
      RdynSweepArgs a;
      memset(&a, 0, sizeof a);
      a.chain = dc;
      a.q = b->q + s0 * in_step;
      a.dq = b->dq + s0 * in_step;
      a.ddq = b->ddq + s0 * in_step;
      a.bcol = tau_meas ? tau_meas + s0 * in_step : nullptr;
      a.bcol_col = cols;
      a.n_samples = cnt;
      rec_strides(b, n, &a.in_ss, &a.in_sj);
      a.Y = img;
      a.y_ss = 1;
      a.y_sr = cnt;
      a.y_sc = (int64_t)n * cnt;
      if (p.long_image)
      {
        // 11 input joints: the dense image by the run-time-length kernel (rdyn_long_local.hip), the measured torque behind it
        RdynLongLocalArgs la;
        memset(&la, 0, sizeof la);
        la.chain_long = dcl;
        la.q = a.q;
        la.dq = a.dq;
        la.ddq = a.ddq;
        la.bcol = a.bcol;
        la.bcol_col = cols;
        la.n_samples = cnt;
        la.in_ss = a.in_ss;
        la.in_sj = a.in_sj;
        la.Y = img;
        la.y_ss = 1;
        la.y_sr = cnt;
        la.y_sc = (int64_t)n * cnt;
        la.n_active = n;
        RDYN_HIP_TRY(rdyn_launch_long_local(RDYN_MODE_REGRESSOR, nJ, la, stream));
      }
      else
        RDYN_HIP_TRY(rdyn_launch_local_sweep(nJ, RDYN_MODE_REGRESSOR_GRAM, a, stream));
      if (K > 0)
      {
        RdynComponentArgs cc = ca;
        cc.q = a.q;
        cc.dq = a.dq;
        cc.n_samples = cnt;
        cc.in_ss = a.in_ss;
        cc.in_sj = a.in_sj;
        cc.n_active = n;
        cc.n_comps = n_comps;
        cc.C = img + (int64_t)10 * nJ * n * cnt;
        cc.c_ss = 1;
        cc.c_sr = cnt;
        cc.c_sc = (int64_t)n * cnt;
        cc.tau = nullptr;
        RDYN_HIP_TRY(rdyn_launch_components(cc, stream));
      }
      // (without measured torques the image's last column stays zero: the factor of [A | 0])
      st = rdyn_tsqr(img, (int64_t)n * cnt, (int64_t)n * cnt, cols, img + (int64_t)cols * n * cnt, R_swept, (s0 > 0 || (accumulate && !expand)) ? 1 : 0,
                     ws + p.rows_off, rdyn_tsqr_workspace_bytes(n1s), -1, stream);
      if (st != RDYN_OK) return st;
    }
    if (expand)
    {
      RdynGramExpandArgs ea;
      memset(&ea, 0, sizeof ea);
      st = device_expand(c, &ea.X);
      if (st != RDYN_OK) return st;
      for (int f = 0; f < c->n_joints(); ++f) ea.red_of[f] = c->red_of[f];
      ea.n_joints = c->n_joints();
      ea.n_red = nJ;
      ea.n_comp_cols = K;
      double* const R_exp = accumulate ? ws + L.r_full : R;
      RDYN_HIP_TRY(rdyn_launch_cholqr_expand(ea, R_swept, R_exp, stream));
      if (accumulate) RDYN_HIP_TRY(rdyn_launch_cholqr_fold(R_exp, R, n1, stream, n1s));
    }
    return RDYN_OK;
  }
  auto bind = [&](RdynLdsGramArgs& la) {
    la.chain = dc;
    la.q = b->q;
    la.dq = b->dq;
    la.ddq = b->ddq;
    la.bcol = tau_meas;
    la.n_samples = b->n_samples;
    rec_strides(b, n, &la.in_ss, &la.in_sj);
    la.n_comps = n_comps;
    la.n_comp_cols = K;
    int col = 0;
    for (int i = 0; i < n_comps; ++i)
    {
      la.comps[i] = ca.comps[i];
      la.comps[i].joint = cs->input_row[ca.comps[i].joint];  // the tile row of the component's input joint (sorted view)
      const int w = ca.comps[i].type == RDYN_COMP_FRICTION2 ? 3 : 2;
      for (int k = 0; k < w; ++k) la.comp_col_row[col++] = (signed char)la.comps[i].joint;
    }
  };
  bind(p.la);
  bind(p.la_b);
  bind(p.la_sub);
  bind(p.la_wide);
  double* const ws = (double*)workspace;
  const int64_t tiles = (b->n_samples + 15) / 16;
  // the Householder folds of all rows: rdyn_tsqr.hip where the factor fits a wave's registers, rdyn_tsqr_wide.hip otherwise
  auto householder = [&](double* R_to, int acc, const int* run_flag, int fan) -> int {
    if (p.nc_reg)
    {
      RdynLdsGramArgs all = p.la;
      all.run_flag = run_flag;
      const int hblocks = (int)((tiles + 3) / 4 < kTsqrBlocks ? (tiles + 3) / 4 : kTsqrBlocks);
      RDYN_HIP_TRY(rdyn_launch_regressor_tsqr(nJ, all, hblocks, 4 * (size_t)p.la.tile_bytes, ws, R_to, acc, stream, fan));
    }
    else
    {
      RdynLdsGramArgs all = p.la_wide;
      all.run_flag = run_flag;
      const int hblocks = (int)(tiles < kTsqrBlocks ? tiles : kTsqrBlocks);
      RDYN_HIP_TRY(rdyn_launch_regressor_tsqr_wide(nJ, all, hblocks, ws, R_to, acc, stream));
    }
    return RDYN_OK;
  };
  // ---- which route.  Preconditioned CholeskyQR (rdyn_cholqr.hip: the heavy pass on the matrix cores) for large batches of chains
  // whose swept form has every joint as an input joint; the Householder folds otherwise.
  const char* route_env = probe_env("RDYN_TSQR_ROUTE");  // A/B builds only: "householder" / "cholqr"
  const int pairs = p.pairs;
  bool cholqr = pairs != 0 && tiles >= kCholqrMinTiles;
  if (route_env && pairs != 0) cholqr = !strcmp(route_env, "cholqr");
  // where the factor of the swept chain goes: straight into R when nothing follows
  double* const R_swept = (expand || (cholqr && accumulate)) ? ws + L.r_swept : R;
  if (cholqr)
  {
    // pass A: the Gram matrix of every S-th tile (about 1 024 tiles whatever the batch size: one tile per wave pair of the regressor ->
    // Gram kernel; 16 384 samples = 115 000 rows for <= 96 columns put the pivots of the second factorisation within a few % of 1).
    // A preconditioner does not have to be a backward-stable factor -- what it is worth is measured on all rows afterwards.
    RdynLdsGramArgs& la = p.la_b;
    RdynLdsGramArgs sub = p.la_sub;
    const int64_t kSubTiles = probe_env("RDYN_CHOLQR_SUBTILES") ? atoll(probe_env("RDYN_CHOLQR_SUBTILES")) : 1024;
    sub.tile_stride = (int)(tiles / kSubTiles > 1 ? tiles / kSubTiles : 1);
    if (sub.tile_stride > 1 && sub.tile_stride % 2 == 0) ++sub.tile_stride;  // odd: does not lock onto power-of-two periods of a trajectory
    const int64_t sub_tiles = (tiles + sub.tile_stride - 1) / sub.tile_stride;
    la.slabs = ws + L.slabs;
    sub.slabs = la.slabs;
    const int np = pairs == -1 ? 4 : (pairs < 0 ? -pairs : pairs);  // tiles a workgroup works on at a time
    const int blocks = (int)((tiles + np - 1) / np < kCholqrBlocks ? (tiles + np - 1) / np : kCholqrBlocks);
    int* const flag = (int*)(ws + L.flag);
    const int col_shift = pairs == -1 ? rdyn_cholqr_solo_col_shift(nJ, K) : rdyn_cholqr_col_shift(nJ, p.xb);
    la.col_shift = col_shift;
    {
      // the plain regressor -> Gram kernel (rdyn_duo_gram.hip) on the subsample: its slab layout (descending link order, component
      // columns in front), its workgroups of four pairs
      const int sub4 = (int)((sub_tiles + 3) / 4 < kCholqrBlocks ? (sub_tiles + 3) / 4 : kCholqrBlocks);
      const int nbt = p.nb;
      size_t lds_bytes = 4 * (size_t)sub.tile_bytes;
      const size_t red_bytes = (size_t)(nbt * (nbt + 1) / 2) * 256 * sizeof(double);
      if (lds_bytes < red_bytes) lds_bytes = red_bytes;
      RDYN_HIP_TRY(rdyn_launch_regressor_gram_duo(10 * nJ, sub, sub4, lds_bytes, stream));
      RdynGramArgs gs;
      memset(&gs, 0, sizeof gs);
      gs.P = n1s - 1;
      gs.slabs = la.slabs;
      gs.G = ws + L.g2;
      gs.c = ws + L.g2 + (size_t)(n1s - 1) * (n1s - 1);
      gs.bb = gs.c + (n1s - 1);
      gs.desc_nj = nJ;
      gs.desc_k = K;
      gs.slab_nb = K > 0 ? nbt : 0;
      RDYN_HIP_TRY(rdyn_launch_gram_finish(gs, sub4, stream));
    }
    st = cholqr_rounds(ws, L, n1s, col_shift, p.nb, K > 0 ? p.nb : 0, tau_meas ? 1 : 0, sqrt((double)tiles / (double)sub_tiles), blocks, R_swept, stream,
                       [&](const double* W, const int* run, double*) -> int {
                         RDYN_HIP_TRY(rdyn_launch_regressor_pgram(nJ, la, W, run, blocks, pairs, stream));
                         return RDYN_OK;
                       });
    if (st != RDYN_OK) return st;
    // stand-by: the Householder factorisation of ALL rows, queued behind the two rounds and started by the device only when a
    // preconditioner was unfit (growth factor) or round 1 was not accepted either (flag[1]; launches that leave at once otherwise).
    // It asks nothing of the batch, so the call as a whole is as robust as the Householder route whatever the subsample looked like.
    st = householder(R_swept, 0, flag + 1, 16);
    if (st != RDYN_OK) return st;
    if (!expand && accumulate) RDYN_HIP_TRY(rdyn_launch_cholqr_fold(R_swept, R, n1s, stream));
  }
  else
  {
    st = householder(R_swept, (accumulate && !expand) ? 1 : 0, nullptr, 2);
    if (st != RDYN_OK) return st;
  }
  if (expand)
  {
    // [A C b] = [A_red C b] diag(E, I_K, 1): R = qr(R_red diag(E, I_K, 1)), folded into the caller's factor when accumulating
    RdynGramExpandArgs ea;
    memset(&ea, 0, sizeof ea);
    st = device_expand(c, &ea.X);
    if (st != RDYN_OK) return st;
    for (int f = 0; f < c->n_joints(); ++f) ea.red_of[f] = c->red_of[f];
    ea.n_joints = c->n_joints();
    ea.n_red = nJ;
    ea.n_comp_cols = K;
    double* const R_exp = accumulate ? ws + L.r_full : R;
    RDYN_HIP_TRY(rdyn_launch_cholqr_expand(ea, R_swept, R_exp, stream));
    if (accumulate) RDYN_HIP_TRY(rdyn_launch_cholqr_fold(R_exp, R, n1, stream, n1s));
  }
  return RDYN_OK;
}

}  // extern "C"
// ---- the pieces of a factor call the multi-device form (rdyn_multi_gpu.cpp) needs separately (C++ linkage, not exported)
__attribute__((visibility("hidden"))) int rdyn_internal_tsqr_widths(const rdyn_chain* c, const rdyn_component* comps, int n_comps, int* n1s, int* n1, int* expands)
{
  TsqrPlan p;
  if (!c || !tsqr_plan(c, comps, n_comps, &p)) return RDYN_ERR_UNSUPPORTED;
  *n1s = p.n1s;
  *n1 = p.n1;
  *expands = p.expand ? 1 : 0;
  return RDYN_OK;
}
__attribute__((visibility("hidden"))) int rdyn_internal_tsqr_swept(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b,
                                                                   const double* tau_meas, double* R_swept, void* workspace, size_t workspace_bytes)
{
  return regressor_tsqr_run(c, comps, n_comps, b, tau_meas, R_swept, 0, workspace, workspace_bytes, "rdyn_identification_tsqr_multi", true);
}
// R <- (accumulate ? qr([R ; .]) : .) of  R_swept diag(E, I_K, 1);  scratch: n1 x n1 doubles (used when accumulating); current device
__attribute__((visibility("hidden"))) int rdyn_internal_tsqr_expand(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const double* R_swept,
                                                                    double* R, int accumulate, double* scratch, void* stream_v)
{
  TsqrPlan p;
  if (!tsqr_plan(c, comps, n_comps, &p) || !p.expand) return RDYN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_v;
  RdynGramExpandArgs ea;
  memset(&ea, 0, sizeof ea);
  int st = device_expand(c, &ea.X);
  if (st != RDYN_OK) return st;
  for (int f = 0; f < c->n_joints(); ++f) ea.red_of[f] = c->red_of[f];
  ea.n_joints = c->n_joints();
  ea.n_red = p.nJ;
  ea.n_comp_cols = p.K;
  double* const R_exp = accumulate ? scratch : R;
  RDYN_HIP_TRY(rdyn_launch_cholqr_expand(ea, R_swept, R_exp, stream));
  if (accumulate) RDYN_HIP_TRY(rdyn_launch_cholqr_fold(R_exp, R, p.n1, stream, p.n1s));
  return RDYN_OK;
}
extern "C"
{

size_t rdyn_regressor_tsqr_workspace_bytes(const rdyn_chain* c)
{
  if (!c) return 0;
  TsqrPlan p;
  return tsqr_plan(c, nullptr, 0, &p) ? p.L.total_doubles * sizeof(double) : 0;
}

int rdyn_regressor_tsqr(const rdyn_chain* c, const rdyn_batch* b, const double* tau_meas, double* R, int accumulate, void* workspace,
                        size_t workspace_bytes)
{
  return regressor_tsqr_run(c, nullptr, 0, b, tau_meas, R, accumulate, workspace, workspace_bytes, "rdyn_regressor_tsqr");
}

size_t rdyn_identification_tsqr_workspace_bytes(const rdyn_chain* c, const rdyn_component* comps, int n_comps)
{
  if (!c || n_comps < 0 || n_comps > RDYN_MAX_COMPONENTS || (n_comps > 0 && !comps)) return 0;
  TsqrPlan p;
  return tsqr_plan(c, comps, n_comps, &p) ? p.L.total_doubles * sizeof(double) : 0;
}

int rdyn_identification_tsqr(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b, const double* tau_meas, double* R,
                             int accumulate, void* workspace, size_t workspace_bytes)
{
  return regressor_tsqr_run(c, comps, n_comps, b, tau_meas, R, accumulate, workspace, workspace_bytes, "rdyn_identification_tsqr");
}

int rdyn_tsqr_last_report(const rdyn_chain* c, const rdyn_component* comps, int n_comps, int64_t n_samples, const void* workspace, int device,
                          void* stream, rdyn_tsqr_report* out)
{
  if (!c || !workspace || !out || n_comps < 0 || n_comps > RDYN_MAX_COMPONENTS || (n_comps > 0 && !comps) || n_samples < 0)
  {
    rdyn_set_error("rdyn_tsqr_last_report: invalid argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  memset(out, 0, sizeof *out);
  TsqrPlan p;
  if (!tsqr_plan(c, comps, n_comps, &p))
  {
    rdyn_set_error("rdyn_tsqr_last_report: the factor entry points do not serve this chain");
    return RDYN_ERR_UNSUPPORTED;
  }
  if (p.image || p.pairs == 0 || (n_samples + 15) / 16 < kCholqrMinTiles) return RDYN_OK;  // route 0: the Householder folds, nothing to report
  DeviceGuard g;
  int st = g.enter(device);
  if (st != RDYN_OK) return st;
  double raw[96];
  RDYN_HIP_TRY(hipMemcpyAsync(raw, (const double*)workspace + p.L.flag, sizeof raw, hipMemcpyDeviceToHost, (hipStream_t)stream));
  RDYN_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  read_report(raw, p.n1s, out);
  return RDYN_OK;
}

// ---- identification step: normal equations of [Y | C | tau_meas] (rigid-body regressor + component columns) -------------
static const int64_t kIdentChunk = 131072;  // samples per image chunk (the image of one chunk stays L2 / Infinity-Cache sized)

size_t rdyn_identification_gram_workspace_bytes(const rdyn_chain* c, const rdyn_component* comps, int n_comps)
{
  if (!c) return 0;
  const int K = n_comps > 0 ? rdyn_components_columns(comps, n_comps) : 0;
  if (K >= 0 && gram_only_through_reduced(c, K))
  {
    const size_t w = c->reduced ? rdyn_identification_gram_workspace_bytes(c->reduced.get(), comps, n_comps) : 0;
    return w ? w + reduce_tmp_bytes(10 * c->reduced->n_joints() + K) : 0;
  }
  const int cols = 10 * c->n_joints() + (K > 0 ? K : 0);
  if (K < 0 || rdyn_gram_blocks_for(cols) > 7) return 0;
  const size_t base = (gram_slab_bytes(cols) + (size_t)kIdentChunk * c->n_active() * (cols + 1) * sizeof(double) + 255) & ~(size_t)255;
  return base + (c->reduced ? reduce_tmp_bytes(10 * c->reduced->n_joints() + (K > 0 ? K : 0)) : 0);
}

int rdyn_identification_gram(const rdyn_chain* c, const rdyn_component* comps, int n_comps, const rdyn_batch* b, const double* tau_meas,
                             double* G, double* cvec, double* bb, int accumulate, void* workspace, size_t workspace_bytes)
{
  int st = check_batch(c, b, true, true, "rdyn_identification_gram", LONG_COMPANION);
  if (st != RDYN_OK) return st;
  if (!G || !workspace || n_comps < 0 || n_comps > RDYN_MAX_COMPONENTS || (n_comps > 0 && !comps))
  {
    rdyn_set_error("rdyn_identification_gram: null output / workspace, or more than %d components", RDYN_MAX_COMPONENTS);
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const int n = c->n_active(), P = 10 * c->n_joints();
  const int K = n_comps > 0 ? rdyn_components_columns(comps, n_comps) : 0;
  const int cols = P + K;
  if (K >= 0 && gram_only_through_reduced(c, K))
  {
    const size_t need = rdyn_identification_gram_workspace_bytes(c, comps, n_comps);
    if (need == 0 || workspace_bytes < need)
    {
      rdyn_set_error("rdyn_identification_gram: %s", need == 0 ? "at most 111 columns (reduced chain + components) are supported" : "workspace too small");
      return need == 0 ? RDYN_ERR_UNSUPPORTED : RDYN_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard g;
    st = g.enter(b->device);
    if (st != RDYN_OK) return st;
    if (b->n_samples == 0)
    {
      if (!accumulate)
      {
        RDYN_HIP_TRY(hipMemsetAsync(G, 0, sizeof(double) * cols * cols, (hipStream_t)b->stream));
        if (cvec) RDYN_HIP_TRY(hipMemsetAsync(cvec, 0, sizeof(double) * cols, (hipStream_t)b->stream));
        if (bb) RDYN_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(double), (hipStream_t)b->stream));
      }
      return RDYN_OK;
    }
    return gram_through_reduced(c, comps, n_comps, K, b, tau_meas, G, cvec, bb, accumulate, workspace, need);
  }
  if (rdyn_gram_blocks_for(cols) > 7)
  {
    rdyn_set_error("rdyn_identification_gram: at most 111 columns (regressor + components) are supported");
    return RDYN_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < rdyn_identification_gram_workspace_bytes(c, comps, n_comps))
  {
    rdyn_set_error("rdyn_identification_gram: workspace too small");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  RdynComponentArgs ca;
  memset(&ca, 0, sizeof ca);
  st = fill_components(comps, n_comps, n, &ca);
  if (st != RDYN_OK) return st;
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  const RdynChainConst* dc = nullptr;
  st = device_const(c, &dc);
  if (st != RDYN_OK) return st;
  hipStream_t stream = (hipStream_t)b->stream;
  double* slabs = (double*)workspace;
  double* scratch = (double*)((char*)workspace + gram_slab_bytes(cols));
  const int64_t N = b->n_samples;
  if (N == 0)
  {
    if (!accumulate)
    {
      RDYN_HIP_TRY(hipMemsetAsync(G, 0, sizeof(double) * cols * cols, stream));
      if (cvec) RDYN_HIP_TRY(hipMemsetAsync(cvec, 0, sizeof(double) * cols, stream));
      if (bb) RDYN_HIP_TRY(hipMemsetAsync(bb, 0, sizeof(double), stream));
    }
    return RDYN_OK;
  }
  if (c->reduced && !probe_env("RDYN_GRAM_NO_REDUCE"))
    return gram_through_reduced(c, comps, n_comps, K, b, tau_meas, G, cvec, bb, accumulate, workspace,
                                rdyn_identification_gram_workspace_bytes(c, comps, n_comps));
  // ---- fused: regressor rows AND component columns stay in LDS (rdyn_duo_gram.hip); else the chunk image below
  if (n >= 2 && n <= 8 && rdyn_regressor_gram_duo_supports_components(P, K) && !probe_env("RDYN_IDENT_UNFUSED"))
  {
    RdynLdsGramArgs la;
    memset(&la, 0, sizeof la);
    const rdyn_chain* const co = ordered(c);  // (input joints out of chain order: the sorted view, rdyn_chain.hpp)
    bool monotonic = build_lds_tile(co, K, false, &la);
    if (4 * (size_t)la.tile_bytes > 160 * 1024) monotonic = build_lds_tile(co, K, false, &la, true);  // compact layout (7 joints + components)
    const int nbt = K > 0 ? rdyn_gram_blocks_for(P) + 1 : rdyn_gram_blocks_for(P);  // the kernel's slab layout (XB = 1 with components)
    size_t lds_bytes = 4 * (size_t)la.tile_bytes;
    if (kin_sweeper_pad(co, K) == 4 && lds_bytes + RDYN_KIN_XCH_BYTES(co->n_joints()) <= 160 * 1024 && la.lds_stride[0] == (16 + 4) * 8)
    {
      la.sweep_lanes = 1;  // one lane per sample: the exchange area behind the tiles
      lds_bytes += RDYN_KIN_XCH_BYTES(co->n_joints());
    }
    const size_t red_bytes = (size_t)(nbt * (nbt + 1) / 2) * 256 * sizeof(double);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    if (monotonic && lds_bytes <= 160 * 1024)
    {
      st = device_const(co, &la.chain);
      if (st != RDYN_OK) return st;
      la.q = b->q;
      la.dq = b->dq;
      la.ddq = b->ddq;
      la.bcol = tau_meas;
      la.n_samples = N;
      rec_strides(b, n, &la.in_ss, &la.in_sj);
      la.slabs = slabs;
      la.n_comps = n_comps;
      la.n_comp_cols = K;
      int col = 0;
      for (int i = 0; i < n_comps; ++i)
      {
        la.comps[i] = ca.comps[i];
        la.comps[i].joint = co->input_row[ca.comps[i].joint];  // the tile row of the component's input joint
        const int w = ca.comps[i].type == RDYN_COMP_FRICTION2 ? 3 : 2;
        for (int k = 0; k < w; ++k) la.comp_col_row[col++] = (signed char)la.comps[i].joint;
      }
      const int64_t tiles = (N + 15) / 16;
      const int blocks = (int)((tiles + 3) / 4 < kFusedBlocks ? (tiles + 3) / 4 : kFusedBlocks);
      RDYN_HIP_TRY(rdyn_launch_regressor_gram_duo(P, la, blocks, lds_bytes, stream));
      RdynGramArgs ga;
      memset(&ga, 0, sizeof ga);
      ga.desc_nj = c->n_joints();  // the kernel accumulates in the order [component columns | tau_meas | links descending]
      ga.desc_k = K;
      ga.slab_nb = nbt;
      ga.P = cols;
      ga.add_to_output = accumulate ? 1 : 0;
      ga.slabs = slabs;
      ga.G = G;
      ga.c = cvec;
      ga.bb = bb;
      RDYN_HIP_TRY(rdyn_launch_gram_finish(ga, blocks, stream));
      return RDYN_OK;
    }
  }
  const int64_t in_step = (b->layout == RDYN_LAYOUT_SAMPLE_MAJOR) ? n : 1;
  int first_col[RDYN_MAX_SWEPT_JOINTS];
  for (int j = 0; j < n; ++j) first_col[j] = 10 * c->active[j];  // the component columns lie to the right: always loaded
  int64_t prev_cnt = -1;
  for (int64_t s0 = 0; s0 < N; s0 += kIdentChunk)
  {
    const int64_t cnt = (N - s0 < kIdentChunk) ? (N - s0) : kIdentChunk;
    if (cnt != prev_cnt)
    {
      RDYN_HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(double) * (size_t)cnt * n * (cols + 1), stream));  // see rdyn_regressor_gram
      prev_cnt = cnt;
    }
    // element-major image of the chunk: rows j * cnt + s, lda = n * cnt; columns [Y (P) | C (K) | tau_meas]
    RdynSweepArgs a;
    memset(&a, 0, sizeof a);
    a.chain = dc;
    a.q = b->q + s0 * in_step;
    a.dq = b->dq + s0 * in_step;
    a.ddq = b->ddq + s0 * in_step;
    a.bcol = tau_meas ? tau_meas + s0 * in_step : nullptr;
    a.bcol_col = cols;
    a.n_samples = cnt;
    rec_strides(b, n, &a.in_ss, &a.in_sj);
    a.Y = scratch;
    a.y_ss = 1;
    a.y_sr = cnt;
    a.y_sc = (int64_t)n * cnt;
    RDYN_HIP_TRY(rdyn_launch_local_sweep(c->n_joints(), RDYN_MODE_REGRESSOR_GRAM, a, stream));
    if (K > 0)
    {
      ca.q = a.q;
      ca.dq = a.dq;
      ca.n_samples = cnt;
      ca.in_ss = a.in_ss;
      ca.in_sj = a.in_sj;
      ca.n_active = n;
      ca.n_comps = n_comps;
      ca.C = scratch + (int64_t)P * n * cnt;
      ca.c_ss = 1;
      ca.c_sr = cnt;
      ca.c_sc = (int64_t)n * cnt;
      ca.tau = nullptr;
      RDYN_HIP_TRY(rdyn_launch_components(ca, stream));
    }
    const bool last = (s0 + cnt >= N);
    st = gram_launch(scratch, (int64_t)n * cnt, (int64_t)n * cnt, cols, tau_meas ? scratch + (int64_t)cols * n * cnt : nullptr, G, cvec, bb,
                     s0 > 0 ? 1 : 0, last, accumulate ? 1 : 0, slabs, stream, cnt, first_col, n);
    if (st != RDYN_OK) return st;
  }
  return RDYN_OK;
}

int rdyn_evaluate_all(const rdyn_chain* c, const rdyn_batch* b, const rdyn_all_outputs* o)
{
  // (a long chain goes through the single-purpose calls below: each one decides for itself what it serves, so that a request for
  // frames / Jacobian / twists / torques of a chain whose inertia and regressor are not served is answered, not refused as a whole)
  int st = check_batch(c, b, true, true, "rdyn_evaluate_all", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!o)
  {
    rdyn_set_error("rdyn_evaluate_all: null outputs");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (b->n_samples == 0) return RDYN_OK;
  const int n = c->n_active(), nJ = c->n_joints(), L = nJ + 1;
  const rdyn_regressor_layout image = {(int64_t)n * 10 * nJ, 1, n};
  const rdyn_regressor_layout* const yl = o->y_layout ? o->y_layout : &image;
  if (c->long_chain())
  {
    // a chain longer than the unrolled kernels sweep: the single-purpose launches (companion + run-time-length kernels), same stream
    if (o->T_links && (st = rdyn_transformation(c, b, nullptr, o->T_links)) != RDYN_OK) return st;
    if (o->J && (st = rdyn_jacobian(c, b, o->J)) != RDYN_OK) return st;
    if ((o->twists || o->dtwists) && (st = rdyn_twist(c, b, o->twists, o->dtwists)) != RDYN_OK) return st;
    if (o->tau && (st = rdyn_joint_torque(c, b, o->tau)) != RDYN_OK) return st;
    if (o->tau_nonlinear && (st = rdyn_joint_torque_nonlinear(c, b, o->tau_nonlinear)) != RDYN_OK) return st;
    if (o->M && (st = rdyn_joint_inertia(c, b, o->M)) != RDYN_OK) return st;
    if (o->Y && (st = rdyn_regressor(c, b, nullptr, o->Y, yl)) != RDYN_OK) return st;
    return RDYN_OK;
  }
  if (o->Y && (yl->stride_sample < 1 || yl->stride_row < 1 || yl->stride_col < 1 || yl->stride_sample > (int64_t)0xFFFFFFFFll / (8 * 255)))
  {
    rdyn_set_error("rdyn_evaluate_all: regressor strides must be positive and stride_sample below %lld doubles", (long long)((int64_t)0xFFFFFFFFll / (8 * 255)));
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard g;
  st = g.enter(b->device);
  if (st != RDYN_OK) return st;
  const RdynChainConst* dc = nullptr;
  st = device_const(c, &dc);
  if (st != RDYN_OK) return st;
  RdynAllArgs a;
  memset(&a, 0, sizeof a);
  a.n_samples = b->n_samples;
  int64_t se;
  auto kin = [&](RdynKinArgs& k) {
    k.chain = dc;
    k.q = b->q;
    k.dq = b->dq;
    k.ddq = b->ddq;
    k.n_samples = b->n_samples;
    rec_strides(b, n, &k.in_ss, &k.in_sj);
    rec_strides(b, 12 * (int64_t)L, &k.tl_ss, &se);
    rec_strides(b, 6 * (int64_t)n, &k.j_ss, &se);
    rec_strides(b, 6 * (int64_t)L, &k.tw_ss, &se);
    k.out_se = se;
    k.j_link = nJ;
  };
  kin(a.frames);
  a.frames.T_links = o->T_links;
  kin(a.jacobian);
  a.jacobian.J = o->J;
  kin(a.twists);
  a.twists.twists = o->twists;
  a.twists.dtwists = o->dtwists;
  auto sweep = [&](RdynSweepArgs& w, bool use_ddq) {
    w.chain = dc;
    w.q = b->q;
    w.dq = b->dq;
    w.ddq = use_ddq ? b->ddq : nullptr;
    w.n_samples = b->n_samples;
    rec_strides(b, n, &w.in_ss, &w.in_sj);
    w.tau_ss = w.in_ss;
    w.tau_sj = w.in_sj;
    rec_strides(b, (int64_t)n * n, &w.m_ss, &w.m_se);
  };
  sweep(a.torque, true);
  a.torque.tau = o->tau;
  sweep(a.torque_nl, false);  // DDq = 0, primitives_impl.h:1287-1288
  a.torque_nl.tau = o->tau_nonlinear;
  sweep(a.inertia, false);
  a.inertia.dq = nullptr;
  a.inertia.M = o->M;
  sweep(a.regressor, true);
  a.regressor.Y = o->Y;
  a.regressor.y_ss = yl->stride_sample;
  a.regressor.y_sr = yl->stride_row;
  a.regressor.y_sc = yl->stride_col;
  RDYN_HIP_TRY(rdyn_launch_sample_all(nJ, a, (hipStream_t)b->stream));
  return RDYN_OK;
}

int rdyn_transformation(const rdyn_chain* c, const rdyn_batch* b, double* T_bt, double* T_links)
{
  int st = check_batch(c, b, false, false, "rdyn_transformation", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!T_bt && !T_links && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_transformation: null outputs");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  return run_base(c, b, T_bt, T_links, nullptr, nullptr, nullptr);
}

int rdyn_jacobian(const rdyn_chain* c, const rdyn_batch* b, double* J)
{
  int st = check_batch(c, b, false, false, "rdyn_jacobian", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!J && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_jacobian: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  return run_base(c, b, nullptr, nullptr, J, nullptr, nullptr);
}

int rdyn_jacobian_link(const rdyn_chain* c, const rdyn_batch* b, int link_index, double* J)
{
  int st = check_batch(c, b, false, false, "rdyn_jacobian_link", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (link_index < 0 || link_index > c->n_joints())
  {
    rdyn_set_error("link index %d is not member of the chain", link_index);  // primitives_impl.h:960
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (!J && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_jacobian_link: null output");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  return run_base(c, b, nullptr, nullptr, J, nullptr, nullptr, link_index);
}

int rdyn_twist(const rdyn_chain* c, const rdyn_batch* b, double* twists, double* dtwists)
{
  int st = check_batch(c, b, true, dtwists != nullptr, "rdyn_twist", LONG_KERNELS);
  if (st != RDYN_OK) return st;
  if (!twists && !dtwists && b->n_samples > 0)
  {
    rdyn_set_error("rdyn_twist: null outputs");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  return run_base(c, b, nullptr, nullptr, nullptr, twists, dtwists);
}

}  // extern "C"
