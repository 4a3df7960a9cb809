// rdyn_image.hip -- dispatcher of the LDS-staged regressor kernels (per-sample drop-in image / stacked matrix).  The kernels live
// in rdyn_image_impl.h and are compiled in slices by number of input joints (rdyn_image_part.hip); this file maps a chain's
// (joints, fixed-joint mask) onto the slice that holds its instantiation.
#include <hip/hip_runtime.h>
#include "rdyn_kernels.h"
#include "rdyn_image_patterns.h"

#define RDYN_IMAGE_NA_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#define RDYN_IMAGE_MULTI_NA_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)

#define DECL(NA_) hipError_t rdyn_image_launch_na##NA_(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* a, hipStream_t st, int mapped);
RDYN_IMAGE_NA_LIST(DECL)
#undef DECL
#define DECL(NA_) \
  hipError_t rdyn_image_launch_multi_na##NA_(int n_joints, unsigned fix, bool stacked, const RdynSweepArgs* table, int n_items, int64_t max_samples, hipStream_t st);
RDYN_IMAGE_MULTI_NA_LIST(DECL)
#undef DECL

namespace
{
inline int popcount_u(unsigned x) { return __builtin_popcount(x); }

// is (n_joints, fix) one of the compiled patterns?
bool pattern_compiled(int n_joints, unsigned fix, int max_na)
{
  if (n_joints < 2 || n_joints > RDYN_MAX_SWEPT_JOINTS || (fix >> n_joints)) return false;
  const int na = n_joints - popcount_u(fix);
  if (na < 2 || na > max_na) return false;
#define MATCH(H_, T_) \
  if (na + H_ + T_ == n_joints && fix == rdyn_image_pattern_mask(na, H_, T_)) return true;
  RDYN_IMAGE_PATTERNS(MATCH)
#undef MATCH
  return false;
}
}  // namespace

// fix_mask: bit f set = chain joint f is not an input joint; the caller has checked that the input joints are in chain order and the
// layout (y_sr == 1, y_sc == n_active; y_ss == n_active selects the stacked matrix)
bool rdyn_image_supported(int n_joints, unsigned fix_mask, int64_t y_ss, bool multi)
{
  if (!pattern_compiled(n_joints, fix_mask, multi ? 8 : RDYN_MAX_SWEPT_JOINTS)) return false;
  const int n_active = n_joints - popcount_u(fix_mask);
  if (y_ss == n_active) return true;  // stacked
  return y_ss > 0 && (y_ss * 8) % 16 == 0 && 64 * y_ss * 8 < (int64_t)0xFFFFFFFFll;
}

// run-time row map (per-sample images only): NA = n_joints - popcount(fix_mask) input joints in any order, the joints of fix_mask anywhere
// the expanded images of a long chain: its reduced companion has 2 .. 6 joints (one flush per link block)
bool rdyn_image_expand_supported(int n_red, int n_full, int64_t y_ss)
{
  return n_red >= 2 && n_red <= 6 && n_full > n_red && n_full <= RDYN_MAX_JOINTS && y_ss >= (int64_t)n_red * 10 * n_full && (y_ss * 8) % 16 == 0 &&
         64 * y_ss * 8 < (int64_t)0xFFFFFFFFll;
}

bool rdyn_image_map_supported(int n_joints, unsigned fix_mask, int64_t y_ss)
{
  if (n_joints < 2 || n_joints > RDYN_IMAGE_MAP_MAX_NJ || (fix_mask >> n_joints)) return false;
  const int nfx = popcount_u(fix_mask), na = n_joints - nfx;
  if (na < 2 || na > RDYN_IMAGE_MAP_MAX_NA || nfx > RDYN_IMAGE_MAP_MAX_FIXED) return false;
  return y_ss >= (int64_t)na * 10 * n_joints && (y_ss * 8) % 16 == 0 && 64 * y_ss * 8 < (int64_t)0xFFFFFFFFll;
}

// mapped = 1: a.row_map[f] = the caller's row of chain joint f, -1 for the joints of fix_mask (k_image_sweep<.., MAP>); a.chain is the
// sorted view when the input joints were listed out of chain order.  mapped = 2: the same map (every joint an input joint) + the blocks
// of a longer chain's links (a.expand_*; k_image_sweep<.., 0, EXPAND>: 2 .. 6 joints); the image stride is the full chain's
hipError_t rdyn_launch_image_sweep(int n_joints, unsigned fix_mask, const RdynSweepArgs& a, hipStream_t st, int mapped)
{
  if (a.n_samples <= 0) return hipSuccess;
  const int n_active = n_joints - popcount_u(fix_mask);
  const bool stacked = a.y_ss == n_active;  // row = s n + j (stacked matrix) instead of one image per sample
  switch (n_active)
  {
#define CASE(NA_) case NA_: return rdyn_image_launch_na##NA_(n_joints, mapped ? 0u : fix_mask, stacked, &a, st, mapped);
    RDYN_IMAGE_NA_LIST(CASE)
#undef CASE
  default: return hipErrorInvalidValue;
  }
}

// every item of `table`: a chain of n_joints joints with the fixed-joint pattern fix_mask, Y in the per-sample image layout
// (stacked == false) or the stacked matrix layout (true); max_samples = largest batch
hipError_t rdyn_launch_image_sweep_multi(int n_joints, unsigned fix_mask, bool stacked, const RdynSweepArgs* table, int n_items, int64_t max_samples,
                                         hipStream_t st)
{
  if (n_items <= 0 || max_samples <= 0) return hipSuccess;
  switch (n_joints - popcount_u(fix_mask))
  {
#define CASE(NA_) case NA_: return rdyn_image_launch_multi_na##NA_(n_joints, fix_mask, stacked, table, n_items, max_samples, st);
    RDYN_IMAGE_MULTI_NA_LIST(CASE)
#undef CASE
  default: return hipErrorInvalidValue;  // longer chains: the plan keeps the strided kernel
  }
}
