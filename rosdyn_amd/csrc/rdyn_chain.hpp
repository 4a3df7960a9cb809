// rdyn_chain.hpp -- host-side chain object behind the opaque `rdyn_chain` handle.
#ifndef RDYN_CHAIN_HPP
#define RDYN_CHAIN_HPP

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rdyn_device.h"

struct rdyn_chain
{
  // what Chain::init keeps (primitives_impl.h:580-703)
  std::vector<rdyn_joint_desc> joints;      // chain order, base -> tool, incl. fixed
  std::vector<rdyn_link_desc> links;        // joints.size() + 1
  std::vector<std::string> moveable_names;  // m_moveable_joints_name
  std::vector<int> active;                  // m_active_joints: chain index per input
  double gravity[3];
  // per chain joint limits (Joint::fromUrdf, primitives_impl.h:85-143)
  std::vector<double> q_max, q_min, dq_max, ddq_max, tau_max;

  std::vector<RdynJointConst> host_joints;  // per chain joint (any number up to RDYN_MAX_JOINTS): constants + child-link parameters
  RdynLongChainConst host_long;             // the same for a chain of more than RDYN_MAX_SWEPT_JOINTS joints (rdyn_long_kin.hip)
  RdynChainConst host_const;                // flat constants for the kernels: valid for chains of <= RDYN_MAX_SWEPT_JOINTS joints only
  // a chain with more joints than the kernels sweep: served through its reduced companion (regressor, torque, inertia, normal
  // equations, R factors); the by-link kinematic outputs are not available for it
  bool long_chain() const { return (int)joints.size() > RDYN_MAX_SWEPT_JOINTS; }

  // lazily created device copies, one per HIP device ordinal; invalidated by set_input_joints
  mutable std::mutex mu;
  mutable std::map<int, RdynChainConst*> dev_const;
  mutable std::map<int, RdynLongChainConst*> dev_long;  // chains of more than RDYN_MAX_SWEPT_JOINTS joints: constants of the run-time-length kernels

  // "Reduced" companion for the regressor -> Gram / factor paths (rdyn_chain_finalize; null when every chain joint is an input joint
  // or the input joints are not in chain order).  Every joint that is not an input joint (fixed, primitives_impl.h:74-83, or left
  // out of setInputJointsName: q = 0) is a constant transform, so the links it connects move as ONE rigid body: the reduced chain
  // keeps the input joints only, with the constant transforms folded into the next input joint's origin.  The ten columns of a
  // link f that was folded away are a constant linear image of the columns of the body it rides on,
  //     Y_f = Y_red(r(f)) X_f,   X_f (10 x 10) = the change of reference frame of the inertial parameters (rigid transform r -> f),
  // and identically zero for links upstream of the first input joint.  Gram / R-factor kernels therefore run on the reduced chain
  // (10 n columns, every joint an input joint: the fastest kernel variants) and a tiny epilogue forms G = E' G_red E.
  std::unique_ptr<rdyn_chain> reduced;
  // "Sorted view" for the kernels that sweep 16-sample tiles into LDS (normal equations, R factors): they number a sample's rows by
  // the CHAIN order of the input joints (the rows a link's columns store are then a prefix).  A'A, A'tau and the R factor do not
  // depend on the order of the rows inside a sample, so a chain whose input joints were listed in another order
  // (setInputJointsName, primitives_impl.h:705-737) is swept through this copy -- same joints, in_idx = rank in chain order -- and
  // the lane that owns row r reads q, Dq, DDq and tau_meas at the caller's input index row_input[r].  Null when the input joints
  // are in chain order already (row_input / input_row are then the identity); chains of <= RDYN_MAX_SWEPT_JOINTS joints only.
  std::unique_ptr<rdyn_chain> sorted;
  std::vector<int> row_input;     // per row of the tile kernels (rank in chain order): the caller's input index
  std::vector<int> input_row;     // the inverse: per input index, its row
  std::vector<int> red_chain;     // per reduced joint: its chain index
  double tail_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, tail_t[3] = {0, 0, 0};  // tool frame of the chain in the companion's tool frame (row-major R)
  std::vector<int> red_of;        // per chain link f + 1 (index f): reduced link it rides on, -1 = upstream of the first input joint
  std::vector<double> expand_X;   // [n_joints][10][10] row-major: X_f(a, p), column p of link f = sum_a Y_red(10 r + a) X_f(a, p)
  mutable std::map<int, double*> dev_expand;  // device copies of expand_X, one per HIP device ordinal

  int n_joints() const { return (int)joints.size(); }
  int n_active() const { return (int)active.size(); }
};

// error plumbing shared by the translation units
void rdyn_set_error(const char* fmt, ...);

// builds host_const from joints/links/active/gravity
void rdyn_chain_finalize(rdyn_chain* c);

// urdf text -> ordered chain description (own minimal XML reader); returns rdyn_status
int rdyn_urdf_extract_chain(const char* xml, const char* base, const char* tool, std::vector<rdyn_joint_desc>& joints,
                            std::vector<rdyn_link_desc>& links);

#endif
