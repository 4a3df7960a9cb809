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

// sin and cos of a joint angle (Joint::computedTpc, primitives_impl.h:38-47: the one transcendental of the path).  ocml's sincos
// carries a Payne-Hanek reduction for arbitrary arguments behind a branch: ~150 instructions where a 750-instruction sample
// spends six of them (profiles/r3/perf_sheet.txt: getTransformation / getJointTorque are ALU-bound).  Joint angles are small:
// for |x| <= 2^20 a two-constant Cody-Waite reduction is exact to ~5e-21 (n = rint(x 2/pi) has <= 20 bits, pio2_1 has 33:
// n pio2_1 is exact, and so is x - n pio2_1 in one fma; the tail constant carries the next 53 bits), followed by the fdlibm kernel
// polynomials on [-pi/4, pi/4] (< 1 ulp) and a quadrant select -- 40 instructions, branch-free.  Beyond 2^20 (or NaN / Inf): ocml.
__device__ __forceinline__ void rdyn_sincos_small(double x, double* sn, double* cs);  // the range-limited part alone: |x| <= 2^20
__device__ __forceinline__ void rdyn_sincos(double x, double* sn, double* cs)
{
  if (__builtin_expect(!(fabs(x) <= 1048576.0), 0))
  {
    sincos(x, sn, cs);
    return;
  }
  rdyn_sincos_small(x, sn, cs);
}
__device__ __forceinline__ void rdyn_sincos_small(double x, double* sn, double* cs)
{
  const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
  const double r = fma(-fn, 1.57079632673412561417e+00, x);  // exact
  const double w = fn * 6.07710050650619224932e-11;
  const double y0 = r - w, y1 = (r - y0) - w;                // reduced argument y0 + y1
  const double z = y0 * y0, v = z * y0;
  // __kernel_sin(y0, y1)
  const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                        8.33333333332248946124e-03);
  const double S = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * -1.66666666666666324348e-01);
  // __kernel_cos(y0, y1)
  const double w2 = z * z;
  const double rc = z * fma(z, fma(z, 2.48015872894767294178e-05, -1.38888888888741095749e-03), 4.16666666666666019037e-02) +
                    (w2 * w2) * fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07);
  const double hz = 0.5 * z, wc = 1.0 - hz;
  const double C = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
  const int n = (int)fn;
  const double s1 = (n & 1) ? C : S, c1 = (n & 1) ? S : C;
  *sn = (n & 2) ? -s1 : s1;
  *cs = ((n + 1) & 2) ? -c1 : c1;
}

}  // namespace

#endif
