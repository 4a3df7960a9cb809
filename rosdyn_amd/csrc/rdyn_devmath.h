// rdyn_devmath.h -- device-side helpers shared by the HIP translation units (3-vectors, constant-address-space access).
#ifndef RDYN_DEVMATH_H
#define RDYN_DEVMATH_H
#include <hip/hip_runtime.h>
#include "rdyn_device.h"

namespace
{

// Chain constants are read through the CONSTANT address space: the loads are then known to be invariant,
// so hipcc emits scalar loads (s_load_dwordx*, counted on lgkmcnt) into SGPRs.  Through a plain global
// pointer the kernel's own Y stores make the compiler fall back to per-lane global_load (counted on
// vmcnt, in order BEHIND the outstanding stores): measured 625 us -> see profiles/r1.
#define RDYN_CONST_AS __attribute__((address_space(4)))
typedef const RDYN_CONST_AS RdynChainConst* ChainPtr;
typedef const RDYN_CONST_AS RdynJointConst& JointRef;
__device__ __forceinline__ ChainPtr as_const(const RdynChainConst* p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (ChainPtr)p;
#pragma clang diagnostic pop
}

struct V3
{
  double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
  return mk(fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x)));
}
// a + b * s
__device__ __forceinline__ V3 axpy(V3 a, V3 b, double s) { return mk(fma(b.x, s, a.x), fma(b.y, s, a.y), fma(b.z, s, a.z)); }
// R^T x, R row-major
__device__ __forceinline__ V3 rotT(const double* R, V3 v)
{
  return mk(fma(R[0], v.x, fma(R[3], v.y, R[6] * v.z)), fma(R[1], v.x, fma(R[4], v.y, R[7] * v.z)),
            fma(R[2], v.x, fma(R[5], v.y, R[8] * v.z)));
}
// R x
__device__ __forceinline__ V3 rot(const double* R, V3 v)
{
  return mk(fma(R[0], v.x, fma(R[1], v.y, R[2] * v.z)), fma(R[3], v.x, fma(R[4], v.y, R[5] * v.z)),
            fma(R[6], v.x, fma(R[7], v.y, R[8] * v.z)));
}
template <class Ptr>
__device__ __forceinline__ V3 ld3(Ptr p) { return mk(p[0], p[1], p[2]); }
// symmetric 3x3 (Ixx Ixy Ixz Iyy Iyz Izz) times vector
template <class Ptr>
__device__ __forceinline__ V3 symv(Ptr I, V3 v)
{
  return mk(fma(I[0], v.x, fma(I[1], v.y, I[2] * v.z)), fma(I[1], v.x, fma(I[3], v.y, I[4] * v.z)),
            fma(I[2], v.x, fma(I[4], v.y, I[5] * v.z)));
}

}  // namespace

#endif
