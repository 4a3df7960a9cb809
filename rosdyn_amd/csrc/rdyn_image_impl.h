// rdyn_image_impl.h -- getRegressor in the DROP-IN layout: every sample's regressor is the contiguous column-major n x P image
// that rosdyn::Chain::getRegressor returns (primitives_impl.h:1350-1354), Y(s, j, p) at s * stride + p * n + j.
//
// k_image_sweep<NJ, FIX>: ONE THREAD PER SAMPLE, the forward local-frame sweep of k_local_sweep (rdyn_kernels.hip: every link
// unrolled, ~46 fp64 instructions per sample and link -- a third of what the row-pair kernels spend).  The ten columns x n rows
// a link contributes to the sample's image (RUN = 80 n contiguous bytes) are NOT stored from the lane that computed them (64
// scattered 8-byte stores per instruction); they go into a per-wave LDS staging area, one ring of RUN + 128 bytes per sample
// addressed by image offset, and after every link the wave writes out, 16 bytes per lane with lanes running along a sample's
// bytes, exactly the WHOLE 128-BYTE LINES of each image that are complete by now; the < 128 bytes behind the last line boundary
// stay in the ring until the next link completes their line.
// Why whole lines: RUN is 3.75 lines at n = 6, so a link-by-link copy-out leaves a partly written line at both ends of every
// run; the two parts arrive a link apart, the L2 has usually evicted the first by then and HBM sees two masked writes
// (read-modify-write under ECC).  Measured on MI355X, N = 1e6, n = 6 / P = 60 (profiles/r2/image_ab.txt): run-by-run copy-out
// 0.86-0.95 ms (slower than the 48-byte row-pair stores it was meant to replace), line-aligned copy-out: see DESIGN.md.
// Only the first / last line of an image can be partial (images are 22.5 lines long): 1 line in 22.
// The ring pitch is RUN + 144 (or 160) bytes, an odd number of 16-byte units: consecutive lanes' 8-byte staging writes fall into
// different LDS banks and the 16-byte reads stay aligned.  Only wave-local ordering is needed (64-thread workgroups, no barrier).  The image stride may be padded
// (stride_sample >= n P); wave bases are 64-bit, per-lane offsets 32-bit inside the wave's 64 images.
//
// Instantiated per FIXED-JOINT PATTERN (round 3): FIX is the bit mask of the chain joints that are not input joints; the input
// joints are the remaining ones, in chain order.  The tile arithmetic folds at compile time for every pattern; the ones compiled
// are "h fixed head joints, NA input joints, t fixed tail joints" with h <= 1, t <= 3 (rdyn_image_patterns.h) -- the reference's
// own test chains in their public URDF form are of that kind: ur10 base_link -> tool0 starts with the fixed joint
// base_link -> base_link_inertia and ends with fixed flange / tool0 frames (rosdyn_speed_test.cpp:44-45, test.cpp:47-48), a Panda
// link0 -> hand ends with two.  Other patterns keep the row-pair kernel (rdyn_rowpair.hip).
// This header is compiled once per number of input joints (rdyn_image_part.hip with -DRDYN_IMAGE_NA=k, -DRDYN_IMAGE_MULTI=0/1) so
// that the translation units build in parallel; rdyn_image.hip holds the dispatcher.
#ifndef RDYN_IMAGE_IMPL_H
#define RDYN_IMAGE_IMPL_H
#include <hip/hip_runtime.h>
#include <atomic>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"
#include "rdyn_image_patterns.h"

namespace
{
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));

__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// pieces a link's ten columns are flushed in (1, 2 or 5).  One piece when a copy-out instruction already covers >= 2 samples
// (NA <= 6: 64 lanes / 8 * ceil(80 NA / 128) chunks); five pieces beyond, where the ring of a whole link leaves 3 waves per CU and
// one sample per store instruction (measured, 1e6 samples, NA = 7: 1078 / 859 / 772 us with 1 / 2 / 5 pieces; NA = 6: 527 / 555 /
// 540 us; NA = 8: 942 / 927 / 945 us -- profiles/r2/image_ab.txt).  -DRDYN_IMAGE_FLUSHES=k forces one value (A/B builds).
constexpr int image_flushes(int na)
{
#ifdef RDYN_IMAGE_FLUSHES
  (void)na;
  return RDYN_IMAGE_FLUSHES;
#else
  return 64 / (((80 * na + 127) / 128) * 8) >= 2 ? 1 : 5;
#endif
}

#ifndef RDYN_IMAGE_WAVES
#define RDYN_IMAGE_WAVES(STACKED_, FIX_, NJ_) ((!(STACKED_) && (FIX_) == 0 && (NJ_) <= 8) ? 2 : 1)
#endif
#ifndef RDYN_STACKED_FLUSHES
#define RDYN_STACKED_FLUSHES 1  // pieces a link's ten columns are staged and copied out in (stacked layout): 1, 2 or 5
#endif
#ifdef RDYN_STACKED_WG256
#define RDYN_IMAGE_WG_WAVES(STACKED_) ((STACKED_) ? 4 : 1)
#else
#define RDYN_IMAGE_WG_WAVES(STACKED_) 1
#endif
// MAP >= 0 (per-sample images; FIX == 0): the input joints of the chain are described at RUN time -- a.row_map[f] = the caller's row of
// chain joint f (where its q / Dq / DDq are read, its torque is written, and the row of the image its values land in), -1 for the MAP
// joints that are not input joints (fixed joints, moving joints left out of setInputJointsName).  Serves input joints listed in any
// order and fixed joints ANYWHERE in the chain (the compiled FIX patterns are heads and tails only); the chain passed is the sorted
// view when there is one (same joints, rows in chain order).
// EXPAND (with MAP = 0): per-sample images of a chain LONGER than the sweep -- `a.chain` is the reduced companion (every joint an input
// joint), the image holds the blocks of all a.expand_n links of the full chain: see rdyn_image_body.inc.
template <int NJ, unsigned FIX, bool NT, bool STACKED, int MAP = -1, bool EXPAND = false>
// (the row-mapped 7-joint chain does not fit the 256 registers of two waves per SIMD: 444 B of scratch, 868 us per 1e6 at 7 joints)
__global__ __launch_bounds__(64 * RDYN_IMAGE_WG_WAVES(STACKED), (RDYN_IMAGE_WG_WAVES(STACKED) > 1 || (MAP == 0 && NJ == 7)) ? 1 : RDYN_IMAGE_WAVES(STACKED, FIX, NJ)) void k_image_sweep(const RdynSweepArgs a)
{
  static_assert(MAP < 0 || (FIX == 0 && !STACKED && MAP < NJ), "row maps: per-sample images, the fixed joints in the map");
  constexpr int IMAGE_WAVES = RDYN_IMAGE_WG_WAVES(STACKED);
#ifdef RDYN_STACKED_XCD_REMAP
  // A/B variant: workgroups are dealt to the 8 XCDs round-robin; give every XCD one contiguous eighth of the batch instead
  const unsigned per = (gridDim.x + 7u) / 8u;
  const unsigned bx = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
#else
  // a.blk_mul != 0: workgroup b sweeps the 64 samples of chunk (b * blk_mul) mod grid (blk_mul coprime to the grid: a bijection) -- the
  // waves that run at the same time are then spread over the whole batch instead of one narrow window of it
  const unsigned bx = a.blk_mul ? (unsigned)(((uint64_t)blockIdx.x * (uint64_t)a.blk_mul) % (uint64_t)gridDim.x) : blockIdx.x;
#endif
  const unsigned blk = bx * IMAGE_WAVES + (IMAGE_WAVES > 1 ? (threadIdx.x >> 6) : 0);
#include "rdyn_image_body.inc"
}

// mixed-chain plan (BASELINE.json configs[4]) in the row-contiguous layouts: blockIdx.y selects one (chain, batch) item of a device
// table; descriptor and chain constants arrive by scalar loads, workgroups past the item's batch leave at once
template <int NJ, unsigned FIX, bool NT, bool STACKED>
__global__ __launch_bounds__(64, RDYN_IMAGE_WAVES(STACKED, FIX, NJ)) void k_image_sweep_multi(const RdynSweepArgs* __restrict__ table)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  const RDYN_CONST_AS RdynSweepArgs& a = *((const RDYN_CONST_AS RdynSweepArgs*)table + blockIdx.y);
#pragma clang diagnostic pop
  constexpr int IMAGE_WAVES = 1;
  constexpr int MAP = -1;
  constexpr bool EXPAND = false;
  const unsigned blk = blockIdx.x;
#include "rdyn_image_body.inc"
}

template <int NJ, unsigned FIX, bool STACKED, int MAP = -1, bool EXPAND = false>
hipError_t launch_image(const RdynSweepArgs& a, hipStream_t st)
{
  constexpr int NA = MAP >= 0 ? NJ - MAP : NJ - __builtin_popcount(FIX);
  constexpr int WV = RDYN_IMAGE_WG_WAVES(STACKED);
  const dim3 grid((unsigned)((a.n_samples + 64 * WV - 1) / (64 * WV)));
  constexpr int runf = (10 / image_flushes(NA)) * NA * 8, w = ((runf + 15) / 16) * 16 + 128, pitch = w + ((w / 16) % 2 ? 32 : 16);
  size_t lds = (STACKED ? (size_t)(10 / RDYN_STACKED_FLUSHES) * 64 * NA * 8 : (size_t)64 * pitch) * WV;
#ifdef RDYN_IMAGE_LDS_FORCE
  lds = RDYN_IMAGE_LDS_FORCE;  // timing experiment: limits the waves per CU through the LDS request
#endif
  // nontemporal copy-out: the lines are written whole, once, and never re-read (A/B, same box: 0.55 ms vs 0.72 ms per 1e6)
#ifdef RDYN_IMAGE_PLAIN_STORES
  constexpr bool kNT = false;
#else
  constexpr bool kNT = true;
#endif
  if constexpr (WV > 1)
  {
    static std::atomic<uint64_t> attr{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!(attr.load() & (1ull << (dev & 63))))
    {
      e = hipFuncSetAttribute((const void*)k_image_sweep<NJ, FIX, kNT, STACKED, MAP, EXPAND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr.fetch_or(1ull << (dev & 63));
    }
  }
  hipLaunchKernelGGL((k_image_sweep<NJ, FIX, kNT, STACKED, MAP, EXPAND>), grid, dim3(64 * WV), lds, st, a);
  return hipGetLastError();
}
template <int NJ, unsigned FIX, bool STACKED>
hipError_t launch_image_multi(const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st)
{
  constexpr int NA = NJ - __builtin_popcount(FIX);
  const dim3 grid((unsigned)((max_samples + 63) / 64), (unsigned)n_items);
  constexpr int runf = (10 / image_flushes(NA)) * NA * 8, w = ((runf + 15) / 16) * 16 + 128, pitch = w + ((w / 16) % 2 ? 32 : 16);
  const size_t lds = STACKED ? (size_t)(10 / RDYN_STACKED_FLUSHES) * 64 * NA * 8 : (size_t)64 * pitch;
  hipLaunchKernelGGL((k_image_sweep_multi<NJ, FIX, true, STACKED>), grid, dim3(64), lds, st, table);
  return hipGetLastError();
}

// one (input joints, fixed head joints, fixed tail joints) pattern; compiled only when the chain fits RDYN_MAX_SWEPT_JOINTS
template <int NA, int H, int T, bool MULTI>
hipError_t image_try(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* a, int n_items, int64_t max_samples, hipStream_t st, bool* hit, bool perm)
{
  if constexpr (NA + H + T <= RDYN_MAX_SWEPT_JOINTS)
  {
    constexpr int NJ = NA + H + T;
    constexpr unsigned FIX = rdyn_image_pattern_mask(NA, H, T);
    if (n_joints == NJ && fix == FIX)
    {
      *hit = true;
      if constexpr (MULTI)
        return stacked ? launch_image_multi<NJ, FIX, true>(a, n_items, max_samples, st) : launch_image_multi<NJ, FIX, false>(a, n_items, max_samples, st);
      else
      {
        if (perm) return hipErrorInvalidValue;  // (row-mapped chains: image_launch_na dispatches them)
        return stacked ? launch_image<NJ, FIX, true>(*a, st) : launch_image<NJ, FIX, false>(*a, st);
      }
    }
  }
  return hipSuccess;
}

template <int NA, bool MULTI>
hipError_t image_launch_na(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* a, int n_items, int64_t max_samples, hipStream_t st, int mapped = 0)
{
  // mapped: 0 = a compiled fixed-joint pattern, 1 = the run-time row map, 2 = the row map + the expansion to a longer chain's links
  if constexpr (!MULTI && image_flushes(NA) == 1)
    if (mapped == 2) return (stacked || fix != 0u || n_joints != NA) ? hipErrorInvalidValue : launch_image<NA, 0u, false, 0, true>(*a, st);
  if (mapped == 2) return hipErrorInvalidValue;
  if constexpr (!MULTI && NA <= RDYN_IMAGE_MAP_MAX_NA)
  {
    // run-time row map: NA input joints among n_joints chain joints (up to RDYN_IMAGE_MAP_MAX_FIXED joints that are not)
    if (mapped)
    {
      if (stacked || fix != 0u) return hipErrorInvalidValue;
      if (n_joints == NA) return launch_image<NA, 0u, false, 0>(*a, st);
      if constexpr (NA + 1 <= RDYN_IMAGE_MAP_MAX_NJ)
        if (n_joints == NA + 1) return launch_image<NA + 1, 0u, false, 1>(*a, st);
      if constexpr (NA + 2 <= RDYN_IMAGE_MAP_MAX_NJ)
        if (n_joints == NA + 2) return launch_image<NA + 2, 0u, false, 2>(*a, st);
      return hipErrorInvalidValue;
    }
  }
  else if (mapped)
    return hipErrorInvalidValue;
  bool hit = false;
  hipError_t e = hipSuccess;
#define RDYN_IMG_TRY(H_, T_) \
  if (!hit) e = image_try<NA, H_, T_, MULTI>(n_joints, fix, stacked, a, n_items, max_samples, st, &hit, false);
  RDYN_IMAGE_PATTERNS(RDYN_IMG_TRY)
#undef RDYN_IMG_TRY
  return hit ? e : hipErrorInvalidValue;
}
}  // namespace
#endif
