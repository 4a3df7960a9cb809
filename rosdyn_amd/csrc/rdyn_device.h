// rdyn_device.h -- chain constants shared by the host ingest (rdyn_chain.cpp) and the HIP kernels.
//
// One RdynChainConst (<= 4 KB) per chain lives in device memory; every kernel reads it through
// wave-uniform addresses, so the compiler turns the reads into scalar loads (s_load_dwordx*) that
// land in SGPRs and broadcast to the 64 lanes for free -- no VGPRs, no LDS bandwidth.  The mixed-chain
// kernels index a table of these by blockIdx.
#ifndef RDYN_DEVICE_H
#define RDYN_DEVICE_H

#include <stdint.h>
#include "../../include/rdyn.h"

// Per chain joint j (child link l = j + 1).  3x3 matrices are row-major.
// Parent->child rotation (reference: Joint::computedTpc, primitives_impl.h:38-47):
//   REVOLUTE : R_pc = R_pj (I + sin q K + (1 - cos q) K^2) = A + sin q * B + (1 - cos q) * C
//   PRISMATIC: R_pc = A, t_pc = t + up * q ;  FIXED: R_pc = A, t_pc = t
struct RdynJointConst
{
  double A[9];   // R_pj                              (primitives_impl.h:68)
  double B[9];   // R_pj * skew(u)                    (primitives_impl.h:70)
  double C[9];   // R_pj * skew(u)^2                  (primitives_impl.h:71)
  double t[3];   // t_pj                              (primitives_impl.h:54)
  double up[3];  // axis in the parent frame R_pj*u   (primitives_impl.h:69)
  double u[3];   // normalised axis in the joint (= child) frame (primitives_impl.h:55-59)
  double pi[10]; // nominal parameters of the child link [m, m c, Ixx Ixy Ixz Iyy Iyz Izz about the origin] (primitives_impl.h:399-417)
  int32_t type;  // rdyn_joint_type
  int32_t in_idx; // index of this joint in the input vectors, -1 if fixed or not an input joint (primitives_impl.h:728)
};

struct RdynChainConst
{
  int32_t n_joints;  // chain joints incl. fixed
  int32_t n_active;  // input joints
  double g[3];       // gravity in the base frame
  RdynJointConst j[RDYN_MAX_SWEPT_JOINTS];
};

// The same for a chain of up to RDYN_MAX_JOINTS joints (12 KB): read by the run-time-length kinematic kernels (rdyn_long_kin.hip), whose
// link loop is rolled -- joint f's constants arrive by scalar loads at a wave-uniform run-time offset.
struct RdynLongChainConst
{
  int32_t n_joints;
  int32_t n_active;
  double g[3];
  RdynJointConst j[RDYN_MAX_JOINTS];
};

#endif
