// rdyn_image.hip -- getRegressor in the DROP-IN layout: every sample's regressor is the contiguous column-major n x P image
// that rosdyn::Chain::getRegressor returns (primitives_impl.h:1350-1354), Y(s, j, p) at s * stride + p * n + j.
//
// k_image_sweep<NJ, NA>: ONE THREAD PER SAMPLE, the forward local-frame sweep of k_local_sweep (rdyn_kernels.hip: every link
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
// Instantiated for the input-joint patterns the tile arithmetic can fold at compile time: the first NA chain joints are the
// input joints 0 .. NA - 1 in order and the remaining NJ - NA joints are fixed (NA = NJ, or NA = NJ - 1: a fixed tool frame).
// Other patterns keep the row-pair kernel (rdyn_rowpair.hip).
#include <hip/hip_runtime.h>
#include "rdyn_device.h"
#include "rdyn_devmath.h"
#include "rdyn_kernels.h"

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

template <int NJ, int NA, bool NT, bool STACKED>
__global__ __launch_bounds__(64, (!STACKED && NJ == NA && NJ <= 8) ? 2 : 1) void k_image_sweep(const RdynSweepArgs a)
{
  const unsigned blk = blockIdx.x;
#include "rdyn_image_body.inc"
}

// mixed-chain plan (BASELINE.json configs[4]) in the row-contiguous layouts: blockIdx.y selects one (chain, batch) item of a device
// table; descriptor and chain constants arrive by scalar loads, workgroups past the item's batch leave at once
template <int NJ, int NA, bool NT, bool STACKED>
__global__ __launch_bounds__(64, (!STACKED && NJ == NA && NJ <= 8) ? 2 : 1) void k_image_sweep_multi(const RdynSweepArgs* __restrict__ table)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  const RDYN_CONST_AS RdynSweepArgs& a = *((const RDYN_CONST_AS RdynSweepArgs*)table + blockIdx.y);
#pragma clang diagnostic pop
  const unsigned blk = blockIdx.x;
#include "rdyn_image_body.inc"
}

template <int NJ, int NA, bool STACKED>
hipError_t launch_image(const RdynSweepArgs& a, hipStream_t st)
{
  const dim3 grid((unsigned)((a.n_samples + 63) / 64));
  constexpr int runf = (10 / image_flushes(NA)) * NA * 8, w = ((runf + 15) / 16) * 16 + 128, pitch = w + ((w / 16) % 2 ? 32 : 16);
  const size_t lds = STACKED ? (size_t)10 * 64 * NA * 8 : (size_t)64 * pitch;
  // nontemporal copy-out: the lines are written whole, once, and never re-read (A/B, same box: 0.55 ms vs 0.72 ms per 1e6)
#ifdef RDYN_IMAGE_PLAIN_STORES
  hipLaunchKernelGGL((k_image_sweep<NJ, NA, false, STACKED>), grid, dim3(64), lds, st, a);
#else
  hipLaunchKernelGGL((k_image_sweep<NJ, NA, true, STACKED>), grid, dim3(64), lds, st, a);
#endif
  return hipGetLastError();
}
template <int NJ, int NA, bool STACKED>
hipError_t launch_image_multi(const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st)
{
  const dim3 grid((unsigned)((max_samples + 63) / 64), (unsigned)n_items);
  constexpr int runf = (10 / image_flushes(NA)) * NA * 8, w = ((runf + 15) / 16) * 16 + 128, pitch = w + ((w / 16) % 2 ? 32 : 16);
  const size_t lds = STACKED ? (size_t)10 * 64 * NA * 8 : (size_t)64 * pitch;
  hipLaunchKernelGGL((k_image_sweep_multi<NJ, NA, true, STACKED>), grid, dim3(64), lds, st, table);
  return hipGetLastError();
}
}  // namespace

// n_fixed_tail = NJ - NA in {0, 1}; the caller has checked the input-joint pattern and the layout (y_sr == 1, y_sc == NA)
bool rdyn_image_supported(int n_joints, int n_active, int64_t y_ss)
{
  if (y_ss == n_active) return n_active >= 2 && n_active <= RDYN_MAX_JOINTS && (n_joints == n_active || n_joints == n_active + 1);  // stacked
  return n_active >= 2 && n_active <= RDYN_MAX_JOINTS && (n_joints == n_active || n_joints == n_active + 1) && y_ss > 0 &&
         (y_ss * 8) % 16 == 0 && 64 * y_ss * 8 < (int64_t)0xFFFFFFFFll;
}

hipError_t rdyn_launch_image_sweep(int n_joints, int n_active, const RdynSweepArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  const bool stacked = a.y_ss == n_active;  // row = s n + j (stacked matrix) instead of one image per sample
#define IMG(NJ_, NA_) \
  if (n_joints == NJ_ && n_active == NA_) return stacked ? launch_image<NJ_, NA_, true>(a, st) : launch_image<NJ_, NA_, false>(a, st);
  IMG(2, 2) IMG(3, 2) IMG(3, 3) IMG(4, 3) IMG(4, 4) IMG(5, 4) IMG(5, 5) IMG(6, 5) IMG(6, 6) IMG(7, 6) IMG(7, 7) IMG(8, 7) IMG(8, 8)
  IMG(9, 8) IMG(9, 9) IMG(10, 9) IMG(10, 10)
#undef IMG
  return hipErrorInvalidValue;
}

// every item of `table`: a chain of n_joints joints whose first n_active joints are the input joints, Y in the per-sample image
// layout (stacked == false) or the stacked matrix layout (true); max_samples = largest batch
hipError_t rdyn_launch_image_sweep_multi(int n_joints, int n_active, bool stacked, const RdynSweepArgs* table, int n_items, int64_t max_samples,
                                         hipStream_t st)
{
  if (n_items <= 0 || max_samples <= 0) return hipSuccess;
#define IMG(NJ_, NA_) \
  if (n_joints == NJ_ && n_active == NA_) \
    return stacked ? launch_image_multi<NJ_, NA_, true>(table, n_items, max_samples, st) : launch_image_multi<NJ_, NA_, false>(table, n_items, max_samples, st);
  IMG(2, 2) IMG(3, 2) IMG(3, 3) IMG(4, 3) IMG(4, 4) IMG(5, 4) IMG(5, 5) IMG(6, 5) IMG(6, 6) IMG(7, 6) IMG(7, 7) IMG(8, 7) IMG(8, 8)
#undef IMG
  return hipErrorInvalidValue;  // longer chains: the plan keeps the strided kernel
}
