// rdyn_chain.hpp -- host-side chain object behind the opaque `rdyn_chain` handle.
#ifndef RDYN_CHAIN_HPP
#define RDYN_CHAIN_HPP

#include <map>
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

  RdynChainConst host_const;                // flat constants for the kernels

  // lazily created device copies, one per HIP device ordinal; invalidated by set_input_joints
  mutable std::mutex mu;
  mutable std::map<int, RdynChainConst*> dev_const;

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
