// rdyn_image_patterns.h -- the fixed-joint patterns the LDS-staged regressor kernels (rdyn_image_impl.h) are compiled for.
// A pattern = H fixed head joints, NA input joints (consecutive, in chain order), T fixed tail joints; "fixed" = not an input joint
// (urdf fixed joints, primitives_impl.h:74-83, and moving joints left out of setInputJointsName, primitives_impl.h:705-829).
#ifndef RDYN_IMAGE_PATTERNS_H
#define RDYN_IMAGE_PATTERNS_H

#define RDYN_IMAGE_MAX_HEAD 1
#define RDYN_IMAGE_MAX_TAIL 3
// X(H, T)
#define RDYN_IMAGE_PATTERNS(X) X(0, 0) X(0, 1) X(0, 2) X(0, 3) X(1, 0) X(1, 1) X(1, 2) X(1, 3)

// bit f set = chain joint f is not an input joint
constexpr unsigned rdyn_image_pattern_mask(int na, int h, int t)
{
  return ((1u << h) - 1u) | (((1u << t) - 1u) << (h + na));
}

// run-time row maps (k_image_sweep<.., MAP>): per-sample images of chains of up to MAX_NJ joints with up to MAX_NA input joints in any
// order and up to MAX_FIXED joints that are not input joints anywhere in the chain
#define RDYN_IMAGE_MAP_MAX_NA 8
#define RDYN_IMAGE_MAP_MAX_FIXED 2
#define RDYN_IMAGE_MAP_MAX_NJ 8  // chain joints (the row-mapped sweep carries one row per chain joint: 9+ joints cost minutes of build time each)

#endif
