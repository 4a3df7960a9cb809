// rosdyn_chain_facade.hpp -- header-only C++ `rosdyn::Chain` over the C-ABI of librdyn_hip.so.
//
// Host-side mirror, in the reference's own language, of the part of `class rosdyn::Chain`
// (rosdyn_core/include/rosdyn_core/primitives.h:235-555) that lies on the accelerated path: same method
// names, same argument meaning, same exceptions.  Every call is evaluated by the HIP kernels (there is no
// CPU fallback); the single-sample methods stage one sample through a pinned buffer -- they exist so that
// code written against rosdyn::Chain compiles and runs unchanged, not for throughput.  For throughput use
// the *Batch methods (device pointers, N samples per call).
//
// With <Eigen/Core> available the signatures are the reference's Eigen types; without it (this build
// image has no Eigen) minimal column-major stand-ins with the same element access are used.
//
// Differences from the reference, all deliberate:
//  * stateless evaluation: no value caches (primitives_impl.h:886, 985, 1088), so the stale-Dq hazard of
//    primitives_impl.h:1111 does not exist; returned references stay valid until the next call of the SAME getter;
//  * construction takes the robot_description XML string (the reference takes a urdf::Model, urdfdom);
//  * setInputJointsName with an unknown name returns false and leaves the chain unchanged.
#ifndef ROSDYN_CHAIN_FACADE_HPP
#define ROSDYN_CHAIN_FACADE_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/rdyn.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define RDYN_FACADE_HAS_EIGEN 1
#endif
#if __has_include(<urdf_model/model.h>)
#include <urdf_model/model.h>  // urdfdom_headers: the createChain(const urdf::ModelInterface&, ...) overload of primitives.h:566
#define RDYN_FACADE_HAS_URDFDOM 1
#endif
#endif

namespace rosdyn
{

#ifdef RDYN_FACADE_HAS_EIGEN
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
using Matrix6Xd = Eigen::Matrix<double, 6, Eigen::Dynamic>;
using Vector6d = Eigen::Matrix<double, 6, 1>;
using Matrix66d = Eigen::Matrix<double, 6, 6>;
using Vector3d = Eigen::Vector3d;
using Affine3d = Eigen::Affine3d;
using VectorOfAffine3d = std::vector<Eigen::Affine3d, Eigen::aligned_allocator<Eigen::Affine3d>>;
using VectorOfVector6d = std::vector<Vector6d, Eigen::aligned_allocator<Vector6d>>;
// the argument type of the component classes: the reference passes const Eigen::Ref<Eigen::VectorXd>& (base_component.h:124-161,
// friction_polynomial1.h:89-133, ideal_spring.h:57-72) -- a writable vector or vector block is bound without a copy
using VectorArg = Eigen::Ref<Eigen::VectorXd>;
namespace detail
{
inline void set_affine(Affine3d& T, const double* m34)  // column-major 3x4 [R | p]
{
  T.setIdentity();
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r) T.matrix()(r, c) = m34[c * 3 + r];
}
}  // namespace detail
#else
// Minimal column-major stand-ins (only what the facade needs).
struct MatrixXd
{
  int r = 0, c = 0;
  std::vector<double> v;
  MatrixXd() {}
  MatrixXd(int rows_, int cols_) : r(rows_), c(cols_), v((size_t)rows_ * cols_, 0.0) {}
  void resize(int rows_, int cols_) { r = rows_; c = cols_; v.assign((size_t)rows_ * cols_, 0.0); }
  int rows() const { return r; }
  int cols() const { return c; }
  double& operator()(int i, int j) { return v[(size_t)j * r + i]; }
  double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
};
struct VectorXd
{
  std::vector<double> v;
  VectorXd() {}
  explicit VectorXd(int n) : v((size_t)n, 0.0) {}
  void resize(int n) { v.assign((size_t)n, 0.0); }
  int rows() const { return (int)v.size(); }
  int size() const { return (int)v.size(); }
  double& operator()(int i) { return v[(size_t)i]; }
  double operator()(int i) const { return v[(size_t)i]; }
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
};
using Matrix6Xd = MatrixXd;
struct Vector6d
{
  double v[6];
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Matrix66d  // column-major 6 x 6
{
  double v[36];
  double& operator()(int i, int j) { return v[j * 6 + i]; }
  double operator()(int i, int j) const { return v[j * 6 + i]; }
  double* data() { return v; }
};
struct Vector3d
{
  double v[3];
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Affine3d  // 3x4 [R | p], column-major
{
  double m[12];
  double operator()(int r_, int c_) const { return m[c_ * 3 + r_]; }
};
using VectorOfAffine3d = std::vector<Affine3d>;
using VectorOfVector6d = std::vector<Vector6d>;
using VectorArg = VectorXd;
namespace detail
{
inline void set_affine(Affine3d& T, const double* m34) { std::memcpy(T.m, m34, sizeof T.m); }
}  // namespace detail
#endif

class Chain;
using ChainPtr = std::shared_ptr<Chain>;
class Link;
class Joint;
using LinkPtr = std::shared_ptr<Link>;
using JointPtr = std::shared_ptr<Joint>;

// Read-only views of the object tree behind a reference Chain (rosdyn::Joint / rosdyn::Link, primitives.h:62-232), as far as the chain
// goes: Chain::getJoints() / getLinks() (primitives.h:265-272).  The reference's objects are mutable and shared between chains; here
// every Chain owns its views and they never change after construction.
class Joint
{
public:
  enum Type { REVOLUTE, PRISMATIC, FIXED };  // primitives.h:66
  const std::string& getName() const { return m_name; }
  const Type& getType() const { return m_type; }
  bool isFixed() const { return m_type == FIXED; }
  const double& getQMax() const { return m_limits[0]; }
  const double& getQMin() const { return m_limits[1]; }
  const double& getDQMax() const { return m_limits[2]; }
  const double& getDDQMax() const { return m_limits[3]; }
  const double& getTauMax() const { return m_limits[4]; }
  LinkPtr getParentLink() const { return m_parent_link.lock(); }
  LinkPtr getChildLink() const { return m_child_link.lock(); }
  // parent <- child at joint value q (Joint::computedTpc, primitives_impl.h:38-47): revolute R_pj (I + sin q K + (1 - cos q) K^2), t_pj;
  // prismatic R_pj, t_pj + R_pj axis q; fixed R_pj, t_pj
  const Affine3d& getTransformation(const double& q = 0)
  {
    double K[9] = {0, -m_axis[2], m_axis[1], m_axis[2], 0, -m_axis[0], -m_axis[1], m_axis[0], 0}, K2[9], M[9], R[9], t[3] = {m_t[0], m_t[1], m_t[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) K2[r * 3 + c] = K[r * 3] * K[c] + K[r * 3 + 1] * K[3 + c] + K[r * 3 + 2] * K[6 + c];
    const double sn = m_type == REVOLUTE ? std::sin(q) : 0.0, oc = m_type == REVOLUTE ? 1.0 - std::cos(q) : 0.0;
    for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0 ? 1.0 : 0.0) + sn * K[i] + oc * K2[i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = m_R[r * 3] * M[c] + m_R[r * 3 + 1] * M[3 + c] + m_R[r * 3 + 2] * M[6 + c];
    if (m_type == PRISMATIC)
      for (int r = 0; r < 3; ++r) t[r] += (m_R[r * 3] * m_axis[0] + m_R[r * 3 + 1] * m_axis[1] + m_R[r * 3 + 2] * m_axis[2]) * q;
    double m34[12];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) m34[c * 3 + r] = R[r * 3 + c];
    for (int r = 0; r < 3; ++r) m34[9 + r] = t[r];
    detail::set_affine(m_last_T_pc, m34);
    return m_last_T_pc;
  }
  // [linear; angular] screw of the child in the parent frame (primitives_impl.h:25-35): revolute [0; R_pj axis], prismatic [R_pj axis; 0]
  const Vector6d& getScrew_of_child_in_parent()
  {
    for (int i = 0; i < 6; ++i) m_screw(i) = 0.0;
    for (int r = 0; r < 3; ++r)
    {
      const double ap = m_R[r * 3] * m_axis[0] + m_R[r * 3 + 1] * m_axis[1] + m_R[r * 3 + 2] * m_axis[2];
      if (m_type == REVOLUTE) m_screw(3 + r) = ap;
      if (m_type == PRISMATIC) m_screw(r) = ap;
    }
    return m_screw;
  }

private:
  friend class Chain;
  std::string m_name;
  Type m_type = FIXED;
  double m_R[9], m_t[3], m_axis[3], m_limits[5];
  std::weak_ptr<Link> m_parent_link, m_child_link;
  Affine3d m_last_T_pc;
  Vector6d m_screw;
};

class Link
{
public:
  const std::string& getName() const { return m_name; }
  const double& getMass() const { return m_mass; }
  const Vector3d& getCog() const { return m_cog; }
  JointPtr getParentJoint() const { return m_parent_joint.lock(); }                    // null for the chain's base link
  std::vector<JointPtr> getChildrenJoints() const { return m_child_joints; }           // inside the chain: at most one
  VectorXd getNominalParameters() const  // primitives_impl.h:399-417
  {
    VectorXd p(10);
    for (int i = 0; i < 10; ++i) p(i) = m_pi[i];
    return p;
  }
  // sum_k pi_k E_k with the reference's ten basis matrices (primitives_impl.h:337-396; spacevect_algebra.h:232-239), linear part first
  const Matrix66d& getSpatialInertia() const { return m_inertia; }
  const std::vector<Matrix66d>& getSpatialInertiaTerms() const { return m_terms; }

private:
  friend class Chain;
  void build()
  {
    m_terms.resize(10);
    for (auto& E : m_terms)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) E(i, j) = 0.0;
    for (int i = 0; i < 3; ++i) m_terms[0](i, i) = 1.0;
    for (int k = 0; k < 3; ++k)  // m c_k: block(3,0) = skew(e_k), block(0,3) = skew(e_k)^T
    {
      double e[3] = {0, 0, 0};
      e[k] = 1.0;
      const double S[3][3] = {{0, -e[2], e[1]}, {e[2], 0, -e[0]}, {-e[1], e[0], 0}};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
        {
          m_terms[1 + k](3 + i, j) = S[i][j];
          m_terms[1 + k](i, 3 + j) = S[j][i];
        }
    }
    const int ij[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int k = 0; k < 6; ++k)
    {
      m_terms[4 + k](3 + ij[k][0], 3 + ij[k][1]) = 1.0;
      m_terms[4 + k](3 + ij[k][1], 3 + ij[k][0]) = 1.0;
    }
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j)
      {
        double acc = 0.0;
        for (int k = 0; k < 10; ++k) acc += m_pi[k] * m_terms[(size_t)k](i, j);
        m_inertia(i, j) = acc;
      }
  }
  std::string m_name;
  double m_mass = 0.0, m_pi[10];
  Vector3d m_cog;
  Matrix66d m_inertia;
  std::vector<Matrix66d> m_terms;
  std::weak_ptr<Joint> m_parent_joint;
  std::vector<JointPtr> m_child_joints;
};

class Chain
{
public:
  // rosdyn::Chain(model, base_link_name, ee_link_name, gravity)   primitives.h:347 -- throws std::runtime_error
  // with the reference's messages ("Base link not found" / "Tool link not found", primitives_impl.h:603, 610).
  Chain(const std::string& robot_description_xml, const std::string& base_link_name, const std::string& ee_link_name,
        const std::array<double, 3>& gravity = {0.0, 0.0, 0.0})
  {
    int st = rdyn_chain_from_urdf(robot_description_xml.c_str(), base_link_name.c_str(), ee_link_name.c_str(), gravity.data(), &m_h);
    if (st != RDYN_OK) throw std::runtime_error(rdyn_last_error());
    refresh();
  }
  // from the flat POD description (what createChain(const urdf::ModelInterface&, ...) below fills from a parsed urdf::Model)
  explicit Chain(const rdyn_chain_desc& desc)
  {
    if (rdyn_chain_from_desc(&desc, &m_h) != RDYN_OK) throw std::runtime_error(rdyn_last_error());
    refresh();
  }
  Chain(const Chain& cpy)  // the reference's copy re-inits from the shared tree; here: an independent clone
  {
    if (rdyn_chain_clone(cpy.m_h, &m_h) != RDYN_OK) throw std::runtime_error(rdyn_last_error());
    refresh();
  }
  Chain& operator=(const Chain& rhs)
  {
    if (this != &rhs)
    {
      release();
      if (rdyn_chain_clone(rhs.m_h, &m_h) != RDYN_OK) throw std::runtime_error(rdyn_last_error());
      refresh();
    }
    return *this;
  }
  Chain(Chain&&) = delete;             // primitives.h:339
  Chain& operator=(Chain&&) = delete;  // primitives.h:341
  ~Chain() { release(); }
  ChainPtr clone() const { return ChainPtr(new Chain(*this)); }  // primitives.h:554

  // ---- getters, primitives.h:362-447
  bool setInputJointsName(const std::vector<std::string>& joints_name)
  {
    std::vector<const char*> p;
    for (auto& s : joints_name) p.push_back(s.c_str());
    int st = rdyn_chain_set_input_joints(m_h, p.data(), (int)p.size());
    if (st == RDYN_ERR_JOINT_NOT_FOUND) return false;  // primitives_impl.h:732-736
    if (st != RDYN_OK) throw std::invalid_argument(rdyn_last_error());
    refresh();
    return true;
  }
  const unsigned int& getLinksNumber() const { return m_links_number; }
  const unsigned int& getJointsNumber() const { return m_joints_number; }
  const unsigned int& getActiveJointsNumber() const { return m_active_joints_number; }
  const std::vector<std::string>& getMoveableJointNames() const { return m_moveable_joints_name; }
  const std::string& getMoveableJointName(const size_t& iAx) const { return m_moveable_joints_name.at(iAx); }
  const std::vector<std::string>& getActiveJointsName() const { return m_active_joints_name; }
  const std::string& getActiveJointName(const size_t& iAx) const { return m_active_joints_name.at(iAx); }
  const std::vector<std::string>& getLinksName() const { return m_links_name; }
  const std::vector<LinkPtr>& getLinks() const { return m_links; }     // primitives.h:265
  const std::vector<JointPtr>& getJoints() const { return m_joints; }  // primitives.h:269 (chain order, fixed joints included)
  const bool& isOk() const { return m_is_chain_ok; }
  const VectorXd& getQMax() const { return m_q_max; }
  const VectorXd& getQMin() const { return m_q_min; }
  const VectorXd& getDQMax() const { return m_Dq_max; }
  const VectorXd& getDDQMax() const { return m_DDq_max; }
  const VectorXd& getTauMax() const { return m_tau_max; }
  std::array<double, 3> getGravity() const
  {
    std::array<double, 3> g;
    rdyn_chain_gravity(m_h, g.data());
    return g;
  }
  VectorXd getNominalParameters()  // primitives.h:548
  {
    VectorXd pi((int)(10 * m_joints_number));
    rdyn_nominal_parameters(m_h, pi.data());
    return pi;
  }
  const rdyn_chain* handle() const { return m_h; }

  // ---- every getter of ONE sample in one round trip (no reference counterpart; the reference caches what a call computed on the way
  // -- m_last_q, primitives_impl.h:886, 985, 1088 -- so its harness, rosdyn_speed_test.cpp:109-185, pays for the frames once per
  // sample).  Here a call through the GPU costs a host round trip (20-30 us) whatever it computes: evaluateAll(q, Dq, DDq) is ONE
  // kernel launch (rdyn_evaluate_all: frames of all links, tool Jacobian, twists + spatial accelerations, joint torque, its non-linear
  // part, joint inertia, regressor side by side) that reads the inputs from and writes its record to pinned host memory -- no copy
  // calls, one synchronisation -- and the getters below answer from that record for as long as they are asked for the same q (Dq,
  // DDq).  A getter called with other inputs evaluates its own function as before and leaves the record alone.
  void evaluateAll(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    const size_t n = m_active_joints_number;
    if ((size_t)q.rows() != n || (size_t)Dq.rows() != n || (size_t)DDq.rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
    ensure_all();
    std::memcpy(m_all_pin, q.data(), n * sizeof(double));
    std::memcpy(m_all_pin + n, Dq.data(), n * sizeof(double));
    std::memcpy(m_all_pin + 2 * n, DDq.data(), n * sizeof(double));
    chk(rdyn_evaluate_all(m_h, &m_all_b, &m_all_out));
    wait_done();
    const double* r = m_all_pin + m_all_in;
    const int L = (int)m_links_number, ni = (int)n, P = (int)(10 * m_joints_number);
    m_all_T_bl.resize(L);
    for (int l = 0; l < L; ++l) detail::set_affine(m_all_T_bl[l], r + m_all_off[0] + 12 * l);
    m_all_jacobian.resize(6, ni);
    std::memcpy(m_all_jacobian.data(), r + m_all_off[1], sizeof(double) * 6 * n);
    m_all_twists.resize(L);
    m_all_Dtwists.resize(L);
    for (int l = 0; l < L; ++l)
      for (int i = 0; i < 6; ++i)
      {
        m_all_twists[l](i) = r[m_all_off[2] + 6 * l + i];
        m_all_Dtwists[l](i) = r[m_all_off[3] + 6 * l + i];
      }
    m_all_tau.resize(ni);
    m_all_tau_nl.resize(ni);
    for (int i = 0; i < ni; ++i)
    {
      m_all_tau(i) = r[m_all_off[4] + i];
      m_all_tau_nl(i) = r[m_all_off[5] + i];
    }
    m_all_inertia.resize(ni, ni);
    std::memcpy(m_all_inertia.data(), r + m_all_off[6], sizeof(double) * n * n);
    m_all_regressor.resize(ni, P);
    std::memcpy(m_all_regressor.data(), r + m_all_off[7], sizeof(double) * n * P);
    m_all_q = q;
    m_all_Dq = Dq;
    m_all_DDq = DDq;
    m_all_valid = true;
  }
  bool isEvaluated(const VectorXd& q) const { return m_all_valid && same(q, m_all_q); }

  // ---- single-sample kinematics / dynamics (primitives.h:452-463, 539-547)
  const Affine3d& getTransformation(const VectorXd& q)
  {
    if (hit(&q, nullptr, nullptr)) return m_all_T_bl.back();
    stage(&q, nullptr, nullptr);
    run(rdyn_transformation(m_h, &m_b, out(0), nullptr), 12);
    detail::set_affine(m_T_bt, m_host.data());
    return m_T_bt;
  }
  const VectorOfAffine3d& getTransformations(const VectorXd& q)
  {
    if (hit(&q, nullptr, nullptr)) return m_all_T_bl;
    stage(&q, nullptr, nullptr);
    run(rdyn_transformation(m_h, &m_b, nullptr, out(0)), 12 * m_links_number);
    m_T_bl.resize(m_links_number);
    for (unsigned int l = 0; l < m_links_number; ++l) detail::set_affine(m_T_bl[l], m_host.data() + 12 * l);
    return m_T_bl;
  }
  const Matrix6Xd& getJacobian(const VectorXd& q)
  {
    if (hit(&q, nullptr, nullptr)) return m_all_jacobian;
    stage(&q, nullptr, nullptr);
    run(rdyn_jacobian(m_h, &m_b, out(0)), 6 * m_active_joints_number);
    m_jacobian.resize(6, (int)m_active_joints_number);
    std::memcpy(m_jacobian.data(), m_host.data(), sizeof(double) * 6 * m_active_joints_number);
    return m_jacobian;
  }
  const VectorOfVector6d& getTwist(const VectorXd& q, const VectorXd& Dq)
  {
    if (hit(&q, &Dq, nullptr)) return m_all_twists;
    stage(&q, &Dq, nullptr);
    run(rdyn_twist(m_h, &m_b, out(0), nullptr), 6 * m_links_number);
    return fill6(m_twists);
  }
  const Vector6d& getTwistTool(const VectorXd& q, const VectorXd& Dq) { return getTwist(q, Dq).back(); }
  // ---- by-name getters (primitives.h:453, 456, 458); same exception type and text as primitives_impl.h:920, 960, 1022
  unsigned int linkIndex(const std::string& link_name) const
  {
    for (unsigned int l = 0; l < m_links_number; ++l)
      if (m_links_name[l] == link_name) return l;
    throw std::invalid_argument("link " + link_name + " is not member of the chain");
  }
  const Affine3d& getTransformationLink(const VectorXd& q, const std::string& link_name)
  {
    const unsigned int l = linkIndex(link_name);
    return getTransformations(q).at(l);
  }
  Matrix6Xd getJacobianLink(const VectorXd& q, const std::string& link_name)
  {
    const unsigned int l = linkIndex(link_name);
    stage(&q, nullptr, nullptr);
    run(rdyn_jacobian_link(m_h, &m_b, (int)l, out(0)), 6 * m_active_joints_number);
    Matrix6Xd jac;
    jac.resize(6, (int)m_active_joints_number);
    std::memcpy(jac.data(), m_host.data(), sizeof(double) * 6 * m_active_joints_number);
    return jac;
  }
  const Vector6d& getTwistLink(const VectorXd& q, const VectorXd& Dq, const std::string& link_name)
  {
    const unsigned int l = linkIndex(link_name);
    return getTwist(q, Dq).at(l);
  }
  const VectorOfVector6d& getDTwist(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    if (hit(&q, &Dq, &DDq)) return m_all_Dtwists;
    stage(&q, &Dq, &DDq);
    run(rdyn_twist(m_h, &m_b, nullptr, out(0)), 6 * m_links_number);
    return fill6(m_Dtwists);
  }
  const Vector6d& getDTwistTool(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq) { return getDTwist(q, Dq, DDq).back(); }
  // split / jerk sweeps, primitives.h:468-488
  const VectorOfVector6d& getDTwistLinearPart(const VectorXd& q, const VectorXd& DDq)
  {
    stage(&q, nullptr, &DDq);
    run(rdyn_twist_parts(m_h, &m_b, nullptr, out(0), nullptr, nullptr), 6 * m_links_number);
    return fill6(m_Dtwists_linear_part);
  }
  const VectorOfVector6d& getDTwistNonLinearPart(const VectorXd& q, const VectorXd& Dq)
  {
    stage(&q, &Dq, nullptr);
    run(rdyn_twist_parts(m_h, &m_b, nullptr, nullptr, out(0), nullptr), 6 * m_links_number);
    return fill6(m_Dtwists_nonlinear_part);
  }
  const VectorOfVector6d& getDDTwist(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq, const VectorXd& DDDq)
  {
    stage(&q, &Dq, &DDq);
    const size_t n = m_active_joints_number;
    if ((size_t)DDDq.rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
    // DDDq travels behind the output area of the staging buffer (the jerk record is 6 L doubles)
    double* d_dddq = out(6 * (size_t)m_links_number);
    std::memcpy(hout(6 * (size_t)m_links_number), DDDq.data(), n * sizeof(double));  // (mapped pinned memory: the kernel reads it in place)
    run(rdyn_twist_parts(m_h, &m_b, d_dddq, nullptr, nullptr, out(0)), 6 * m_links_number);
    return fill6(m_DDtwists);
  }
  // jerk split, primitives.h:476-484
  const VectorOfVector6d& getDDTwistLinearPart(const VectorXd& q, const VectorXd& DDDq)
  {
    stage(&q, nullptr, nullptr);
    if ((size_t)DDDq.rows() != m_active_joints_number) throw std::invalid_argument("Input data dimensions mismatch");
    double* d_dddq = out(6 * (size_t)m_links_number);
    std::memcpy(hout(6 * (size_t)m_links_number), DDDq.data(), m_active_joints_number * sizeof(double));
    run(rdyn_jerk_parts(m_h, &m_b, d_dddq, out(0), nullptr), 6 * m_links_number);
    return fill6(m_DDtwists_linear_part);
  }
  const VectorOfVector6d& getDDTwistNonLinearPart(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    stage(&q, &Dq, &DDq);
    run(rdyn_jerk_parts(m_h, &m_b, nullptr, nullptr, out(0)), 6 * m_links_number);
    return fill6(m_DDtwists_nonlinear_part);
  }
  const Vector6d& getDDTwistLinearPartTool(const VectorXd& q, const VectorXd& DDDq) { return getDDTwistLinearPart(q, DDDq).back(); }
  const Vector6d& getDDTwistNonLinearPartTool(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    return getDDTwistNonLinearPart(q, Dq, DDq).back();
  }
  // link wrenches, primitives.h:530-535 (base-frame coordinates, referred to each link's origin)
  const VectorOfVector6d& getWrench(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq, const VectorOfVector6d& ext_wrenches_in_link_frame)
  {
    if (ext_wrenches_in_link_frame.size() != m_links_number) throw std::invalid_argument("Input data dimensions mismatch");
    stage(&q, &Dq, &DDq);
    double* const e = hout(6 * (size_t)m_links_number);  // behind the wrench record, in the mapped staging buffer
    for (unsigned l = 0; l < m_links_number; ++l)
      for (int i = 0; i < 6; ++i) e[6 * l + i] = ext_wrenches_in_link_frame[l](i);
    double* d_ext = out(6 * (size_t)m_links_number);
    run(rdyn_wrench(m_h, &m_b, d_ext, out(0)), 6 * m_links_number);
    return fill6(m_wrenches);
  }
  const Vector6d& getWrenchTool(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq, const VectorOfVector6d& ext_wrenches_in_link_frame)
  {
    return getWrench(q, Dq, DDq, ext_wrenches_in_link_frame).back();
  }
  // per-axis limit getters and the input-joint lookup, primitives.h:434-447
  double getQMax(int iAx) const { return m_q_max(iAx); }
  double getQMin(int iAx) const { return m_q_min(iAx); }
  double getDQMax(int iAx) const { return m_Dq_max(iAx); }
  double getDDQMax(int iAx) const { return m_DDq_max(iAx); }
  double getTauMax(int iAx) const { return m_tau_max(iAx); }
  int jointIndex(const std::string& name) const
  {
    for (unsigned i = 0; i < m_active_joints_number; ++i)
      if (m_active_joints_name[i] == name) return (int)i;
    return -1;
  }
  // tool-link shortcuts (primitives.h:459-488), used by the reference's harness (rosdyn_speed_test.cpp:187-191)
  const Vector6d& getDTwistLinearPartTool(const VectorXd& q, const VectorXd& DDq) { return getDTwistLinearPart(q, DDq).back(); }
  const Vector6d& getDTwistNonLinearPartTool(const VectorXd& q, const VectorXd& Dq) { return getDTwistNonLinearPart(q, Dq).back(); }
  const Vector6d& getDDTwistTool(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq, const VectorXd& DDDq)
  {
    return getDDTwist(q, Dq, DDq, DDDq).back();
  }
  // getJointTorque with external wrenches applied TO the links, in link frames (primitives.h:539)
  const VectorXd& getJointTorque(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq, const VectorOfVector6d& ext_wrenches_in_link_frame)
  {
    if (ext_wrenches_in_link_frame.size() != m_links_number) throw std::invalid_argument("Input data dimensions mismatch");
    stage(&q, &Dq, &DDq);
    double* const e = hout(m_active_joints_number);  // behind the torque record, in the mapped staging buffer
    for (unsigned l = 0; l < m_links_number; ++l)
      for (int i = 0; i < 6; ++i) e[6 * l + i] = ext_wrenches_in_link_frame[l](i);
    double* d_ext = out(m_active_joints_number);
    run(rdyn_joint_torque_ext(m_h, &m_b, d_ext, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  const VectorXd& getJointTorque(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    if (hit(&q, &Dq, &DDq)) return m_all_tau;
    stage(&q, &Dq, &DDq);
    run(rdyn_joint_torque(m_h, &m_b, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  const VectorXd& getJointTorqueNonLinearPart(const VectorXd& q, const VectorXd& Dq)
  {
    if (hit(&q, &Dq, nullptr)) return m_all_tau_nl;
    stage(&q, &Dq, nullptr);
    run(rdyn_joint_torque_nonlinear(m_h, &m_b, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  MatrixXd getRegressor(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)  // by value, primitives.h:543
  {
    if (q.rows() != Dq.rows() || Dq.rows() != DDq.rows())
      throw std::invalid_argument("Input data dimensions mismatch");  // primitives_impl.h:1299-1309
    if (hit(&q, &Dq, &DDq)) return m_all_regressor;
    stage(&q, &Dq, &DDq);
    const int n = (int)m_active_joints_number, P = (int)(10 * m_joints_number);
    rdyn_regressor_layout yl = {(int64_t)n * P, 1, n};
    run(rdyn_regressor(m_h, &m_b, nullptr, out(0), &yl), (size_t)n * P);
    MatrixXd Y(n, P);
    std::memcpy(Y.data(), m_host.data(), sizeof(double) * n * P);
    return Y;
  }
  const MatrixXd& getJointInertia(const VectorXd& q)
  {
    if (hit(&q, nullptr, nullptr)) return m_all_inertia;
    stage(&q, nullptr, nullptr);
    const int n = (int)m_active_joints_number;
    run(rdyn_joint_inertia(m_h, &m_b, out(0)), (size_t)n * n);
    m_joint_inertia.resize(n, n);
    std::memcpy(m_joint_inertia.data(), m_host.data(), sizeof(double) * n * n);
    return m_joint_inertia;
  }

  // ---- local inverse kinematics (primitives.h:510, 526).  The reference's wall-clock budget `max_time` becomes an
  // iteration cap; returns the reference's bool (false also when the QP of an iterate is not positive definite).
  bool computeLocalIk(VectorXd& sol, const Affine3d& T_b_t, const VectorXd& seed, const double& toll = 1e-4, int max_iterations = 100)
  {
    return localIk(sol, T_b_t, nullptr, seed, toll, max_iterations);
  }
  bool computeWeigthedLocalIk(VectorXd& sol, const Affine3d& T_b_t, const Vector6d& weight, const VectorXd& seed, const double& toll = 1e-4,
                              int max_iterations = 100)
  {
    double w[6];
    for (int i = 0; i < 6; ++i) w[i] = weight(i);
    return localIk(sol, T_b_t, w, seed, toll, max_iterations);
  }
  void computeLocalIkBatch(const rdyn_batch& seeds, const double* T_target, const double* weight, double toll, int max_iterations, double* sol,
                           int32_t* status, int32_t* iterations) const
  {
    chk(rdyn_local_ik(m_h, &seeds, T_target, weight, toll, max_iterations, sol, status, iterations));
  }

  // ---- getMultiplicity, primitives.h:552 / primitives_impl.h:1470-1516: every joint vector equal to q up to whole turns of the
  // revolute input joints inside [q_min, q_max] (host only).  The reference enumerates up to its 1e10 default limits; a joint
  // without finite limits is refused instead of exhausting memory.
  std::vector<VectorXd> getMultiplicity(const VectorXd& q) const
  {
    const unsigned n = m_active_joints_number;
    if ((unsigned)q.rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<std::vector<double>> multiturn_ax(n);
    for (unsigned idx = 0; idx < n; ++idx)
    {
      multiturn_ax[idx].push_back(q((int)idx));
      if (!m_active_is_revolute[idx]) continue;
      if (m_q_max((int)idx) - m_q_min((int)idx) > two_pi * 1e4)
        throw std::invalid_argument("getMultiplicity: joint " + m_active_joints_name[idx] + " has no finite position limits");
      for (double tmp = q((int)idx) + two_pi; !(tmp > m_q_max((int)idx)); tmp += two_pi) multiturn_ax[idx].push_back(tmp);
      for (double tmp = q((int)idx) - two_pi; !(tmp < m_q_min((int)idx)); tmp -= two_pi) multiturn_ax[idx].push_back(tmp);
    }
    std::vector<VectorXd> multiturn;
    multiturn.push_back(q);
    for (unsigned idx = 0; idx < n; ++idx)
    {
      const size_t size_multiturn = multiturn.size();
      for (size_t is = 1; is < multiturn_ax[idx].size(); ++is)
        for (size_t im = 0; im < size_multiturn; ++im)
        {
          VectorXd new_q = multiturn[im];
          new_q((int)idx) = multiturn_ax[idx][is];
          multiturn.push_back(new_q);
        }
    }
    return multiturn;
  }

  // ---- normal equations of the stacked regressor on the fp64 matrix cores (no counterpart inside rosdyn_core: the consumer of
  // getRegressor was the external rosdyn_identification, README.md:15); device pointers, see include/rdyn.h
  size_t getRegressorGramWorkspaceBytes() const { return rdyn_regressor_gram_workspace_bytes(m_h, 0); }
  void getRegressorGramBatch(const rdyn_batch& b, const double* tau_meas, double* G, double* c, double* bb, bool accumulate, void* workspace,
                             size_t workspace_bytes) const
  {
    chk(rdyn_regressor_gram(m_h, &b, tau_meas, G, c, bb, accumulate ? 1 : 0, 0, workspace, workspace_bytes));
  }
  size_t getIdentificationGramWorkspaceBytes(const std::vector<rdyn_component>& comps) const
  {
    return rdyn_identification_gram_workspace_bytes(m_h, comps.data(), (int)comps.size());
  }
  // [Y | C] with the columns of the per-joint components (ComponentBase::descriptor() of each, below)
  void getIdentificationGramBatch(const std::vector<rdyn_component>& comps, const rdyn_batch& b, const double* tau_meas, double* G, double* c,
                                  double* bb, bool accumulate, void* workspace, size_t workspace_bytes) const
  {
    chk(rdyn_identification_gram(m_h, comps.data(), (int)comps.size(), &b, tau_meas, G, c, bb, accumulate ? 1 : 0, workspace, workspace_bytes));
  }
  // the same identification step WITHOUT forming the normal equations: R1 = [R d; 0 rho] of [Y | C | tau_meas] by tall-skinny QR
  // (condition number not squared; include/rdyn.h: rdyn_regressor_tsqr / rdyn_identification_tsqr), solved by solveRFactor
  size_t getIdentificationTsqrWorkspaceBytes(const std::vector<rdyn_component>& comps) const
  {
    return rdyn_identification_tsqr_workspace_bytes(m_h, comps.data(), (int)comps.size());
  }
  void getIdentificationTsqrBatch(const std::vector<rdyn_component>& comps, const rdyn_batch& b, const double* tau_meas, double* R1, bool accumulate,
                                  void* workspace, size_t workspace_bytes) const
  {
    chk(rdyn_identification_tsqr(m_h, comps.data(), (int)comps.size(), &b, tau_meas, R1, accumulate ? 1 : 0, workspace, workspace_bytes));
  }
  // the R factor of the stacked [regressor | tau_meas] of a batch without the normal equations (include/rdyn.h: rdyn_regressor_tsqr:
  // Householder folds, from 4 096 samples on preconditioned CholeskyQR on the matrix cores); R1 = (10 joints + 1)^2 doubles, device
  size_t getRegressorTsqrWorkspaceBytes() const { return rdyn_regressor_tsqr_workspace_bytes(m_h); }
  void getRegressorTsqrBatch(const rdyn_batch& b, const double* tau_meas, double* R1, bool accumulate, void* workspace, size_t workspace_bytes) const
  {
    chk(rdyn_regressor_tsqr(m_h, &b, tau_meas, R1, accumulate ? 1 : 0, workspace, workspace_bytes));
  }
  // what the last factor call on `workspace` did (include/rdyn.h: rdyn_tsqr_last_report): route, the stage of the preconditioned route that
  // vouched for the result (0 / 1: first / second round, 2: the stand-by Householder factorisation), growth factor and conditioning
  rdyn_tsqr_report getTsqrReport(int64_t n_samples, const void* workspace, int device = -1, void* stream = nullptr,
                                 const std::vector<rdyn_component>& comps = std::vector<rdyn_component>()) const
  {
    rdyn_tsqr_report rep;
    chk(rdyn_tsqr_last_report(m_h, comps.empty() ? nullptr : comps.data(), (int)comps.size(), n_samples, workspace, device, stream, &rep));
    return rep;
  }
  // rigid-body reduction of a chain whose input joints are a subset of its joints (include/rdyn.h: rdyn_chain_reduction): returns the
  // number of bodies (0: no reduction); body_joint[f] = chain index of the input joint link f + 1 rides on (-1: on the base),
  // X = [joints][10][10] row-major with Y(:, 10 f + p) = sum_a Y(:, 10 body_joint[f] + a) X[f][a][p], pi_body = merged parameters
  int getBodyReduction(std::vector<int32_t>& body_joint, std::vector<double>& X, VectorXd& pi_body) const
  {
    const int nb = rdyn_chain_reduction(m_h, nullptr, nullptr, nullptr);
    if (nb <= 0) return 0;
    const int nj = rdyn_chain_joints_number(m_h);
    body_joint.assign((size_t)nj, -1);
    X.assign((size_t)nj * 100, 0.0);
    pi_body.resize(10 * nb);
    rdyn_chain_reduction(m_h, body_joint.data(), X.data(), pi_body.data());
    return nb;
  }
  // minimum-norm solution from a factor R1 ((n + 1) x (n + 1), column-major, copied back to the host); returns the numerical rank
  static int solveRFactor(const MatrixXd& R1, VectorXd& x, double rtol = 1e-10)
  {
    const int n = (int)R1.rows() - 1;
    if (n < 1 || R1.cols() != n + 1) throw std::invalid_argument("Input data dimensions mismatch");
    x.resize(n);
    int rank = 0;
    chk(rdyn_solve_r_factor(R1.data(), n + 1, n, n, R1.data() + (size_t)n * (n + 1), rtol, x.data(), &rank));
    return rank;
  }
  // minimum-norm base-parameter solve of G x = c on the host (G, c already copied back); returns the numerical rank
  static int solveNormalEquations(const MatrixXd& G, const VectorXd& c, VectorXd& x, double rtol = 1e-10)
  {
    const int n = (int)c.rows();
    if (G.rows() != n || G.cols() != n) throw std::invalid_argument("Input data dimensions mismatch");
    x.resize(n);
    int rank = 0;
    chk(rdyn_solve_normal_equations(G.data(), c.data(), n, rtol, x.data(), &rank));
    return rank;
  }

  // ---- batched evaluation on device pointers (what the kernels are for); see include/rdyn.h for layouts
  void getJointTorqueBatch(const rdyn_batch& b, double* tau) const { chk(rdyn_joint_torque(m_h, &b, tau)); }
  void getRegressorBatch(const rdyn_batch& b, double* tau, double* Y, const rdyn_regressor_layout& yl) const
  {
    chk(rdyn_regressor(m_h, &b, tau, Y, &yl));
  }
  void getJointInertiaBatch(const rdyn_batch& b, double* M) const { chk(rdyn_joint_inertia(m_h, &b, M)); }
  void getTransformationBatch(const rdyn_batch& b, double* T_bt, double* T_links) const { chk(rdyn_transformation(m_h, &b, T_bt, T_links)); }
  void getJacobianBatch(const rdyn_batch& b, double* J) const { chk(rdyn_jacobian(m_h, &b, J)); }
  void getTwistBatch(const rdyn_batch& b, double* twists, double* dtwists) const { chk(rdyn_twist(m_h, &b, twists, dtwists)); }

private:
  rdyn_chain* m_h = nullptr;
  unsigned int m_links_number = 0, m_joints_number = 0, m_active_joints_number = 0;
  bool m_is_chain_ok = true;
  std::vector<std::string> m_links_name, m_moveable_joints_name, m_active_joints_name;
  std::vector<char> m_active_is_revolute;  // per input joint (getMultiplicity)
  VectorXd m_q_max, m_q_min, m_Dq_max, m_DDq_max, m_tau_max, m_active_joint_torques;
  std::vector<LinkPtr> m_links;
  std::vector<JointPtr> m_joints;
  Affine3d m_T_bt;
  VectorOfAffine3d m_T_bl;
  Matrix6Xd m_jacobian;
  VectorOfVector6d m_twists, m_Dtwists, m_Dtwists_linear_part, m_Dtwists_nonlinear_part, m_DDtwists, m_DDtwists_linear_part,
      m_DDtwists_nonlinear_part, m_wrenches;
  MatrixXd m_joint_inertia;
  // evaluateAll: its pinned host record (inputs | outputs; the device reads and writes it in place), what it brought back
  double* m_all_pin = nullptr;
  double* m_all_devptr = nullptr;  // the device's address of the same memory
  rdyn_batch m_all_b;
  rdyn_all_outputs m_all_out;
  rdyn_regressor_layout m_all_yl;
  size_t m_all_in = 0;                                 // doubles the inputs take at the head of both records (3 n, rounded to 16 bytes)
  size_t m_all_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // T_links | J | twists | dtwists | tau | tau_nl | M | Y | end (doubles, behind the inputs)
  bool m_all_valid = false;
  VectorXd m_all_q, m_all_Dq, m_all_DDq, m_all_tau, m_all_tau_nl;
  VectorOfAffine3d m_all_T_bl;
  Matrix6Xd m_all_jacobian;
  VectorOfVector6d m_all_twists, m_all_Dtwists;
  MatrixXd m_all_inertia, m_all_regressor;
  // staging for ONE sample: MAPPED pinned host memory (m_dev = the device's address of m_pin) -- the kernels read their inputs from and
  // write their records to host memory directly, a call is one launch and one synchronisation (round 5 paid two hipMemcpyAsync on top:
  // 20-35 us per getter against 10 us for the launch itself)
  double* m_dev = nullptr;
  double* m_pin = nullptr;
  size_t m_dev_doubles = 0, m_need_doubles = 0;
  // completion word of the single-sample calls (mapped pinned memory): a stream memory operation behind the launch writes the call's
  // sequence number, the host spins on it -- 3 us less than hipStreamSynchronize on this stack (tools/latency_probe.hip: 21.5 -> 18.3 us
  // for a small kernel on mapped memory; the driver's own wait is 9.7 us for an EMPTY kernel); falls back to the synchronisation when the
  // stream operation is not available or the word does not arrive
  uint32_t* m_done_pin = nullptr;
  uint32_t* m_done_dev = nullptr;
  uint32_t m_done_seq = 0;
  bool m_done_ok = true;
  std::vector<double> m_host;
  rdyn_batch m_b;

  static void chk(int st)
  {
    if (st == RDYN_ERR_INVALID_ARGUMENT) throw std::invalid_argument(rdyn_last_error());
    if (st != RDYN_OK) throw std::runtime_error(rdyn_last_error());
  }
  static void hip(hipError_t e)
  {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
  }
  static bool same(const VectorXd& a, const VectorXd& b)
  {
    if (a.rows() != b.rows()) return false;
    for (int i = 0; i < (int)a.rows(); ++i)
      if (!(a(i) == b(i))) return false;
    return true;
  }
  // the getter is asked for the sample evaluateAll holds (the arguments it does not take are not compared)
  bool hit(const VectorXd* q, const VectorXd* dq, const VectorXd* ddq) const
  {
    return m_all_valid && same(*q, m_all_q) && (!dq || same(*dq, m_all_Dq)) && (!ddq || same(*ddq, m_all_DDq));
  }
  void release_all()
  {
    m_all_valid = false;
    if (m_all_pin) (void)hipHostFree(m_all_pin);
    m_all_pin = m_all_devptr = nullptr;
  }
  void ensure_all()
  {
    if (m_all_pin) return;
    const size_t n = m_active_joints_number, L = m_links_number, P = 10 * (size_t)m_joints_number;
    const size_t sizes[8] = {12 * L, 6 * n, 6 * L, 6 * L, n, n, n * n, n * P};
    size_t off = 0;
    for (int k = 0; k < 8; ++k)
    {
      m_all_off[k] = off;
      off += (sizes[k] + 1) & ~(size_t)1;
    }
    m_all_off[8] = off;
    m_all_in = (3 * n + 1) & ~(size_t)1;
    hip(hipHostMalloc((void**)&m_all_pin, (m_all_in + off) * sizeof(double), hipHostMallocMapped));
    std::memset(m_all_pin, 0, (m_all_in + off) * sizeof(double));
    if (hipHostGetDevicePointer((void**)&m_all_devptr, m_all_pin, 0) != hipSuccess)
    {
      release_all();
      throw std::runtime_error("HIP: pinned host memory is not addressable by the device");
    }
    std::memset(&m_all_b, 0, sizeof m_all_b);
    m_all_b.n_samples = 1;
    m_all_b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
    m_all_b.device = -1;
    m_all_b.q = m_all_devptr;
    m_all_b.dq = m_all_devptr + n;
    m_all_b.ddq = m_all_devptr + 2 * n;
    double* const o = m_all_devptr + m_all_in;
    m_all_yl = {(int64_t)(n * P), 1, (int64_t)n};
    m_all_out.T_links = o + m_all_off[0];
    m_all_out.J = o + m_all_off[1];
    m_all_out.twists = o + m_all_off[2];
    m_all_out.dtwists = o + m_all_off[3];
    m_all_out.tau = o + m_all_off[4];
    m_all_out.tau_nonlinear = o + m_all_off[5];
    m_all_out.M = o + m_all_off[6];
    m_all_out.Y = o + m_all_off[7];
    m_all_out.y_layout = &m_all_yl;
  }
  void release()
  {
    release_all();
    if (m_pin) (void)hipHostFree(m_pin);
    m_dev = m_pin = nullptr;
    if (m_done_pin) (void)hipHostFree(m_done_pin);
    m_done_pin = m_done_dev = nullptr;
    m_dev_doubles = 0;  // refresh() must allocate again (copy-assignment: release, clone, refresh)
    if (m_h) rdyn_chain_destroy(m_h);
    m_h = nullptr;
  }
  void refresh()
  {
    release_all();  // (setInputJointsName: other sizes)
    m_links_number = (unsigned)rdyn_chain_links_number(m_h);
    m_joints_number = (unsigned)rdyn_chain_joints_number(m_h);
    m_active_joints_number = (unsigned)rdyn_chain_active_joints_number(m_h);
    m_links_name.clear();
    m_moveable_joints_name.clear();
    m_active_joints_name.clear();
    for (unsigned i = 0; i < m_links_number; ++i) m_links_name.push_back(rdyn_chain_link_name(m_h, (int)i));
    for (int i = 0; i < rdyn_chain_moveable_joints_number(m_h); ++i) m_moveable_joints_name.push_back(rdyn_chain_moveable_joint_name(m_h, i));
    for (unsigned i = 0; i < m_active_joints_number; ++i) m_active_joints_name.push_back(rdyn_chain_active_joint_name(m_h, (int)i));
    m_active_is_revolute.assign(m_active_joints_number, 0);
    for (unsigned i = 0; i < m_active_joints_number; ++i)
      for (unsigned j = 0; j < m_joints_number; ++j)
        if (m_active_joints_name[i] == rdyn_chain_joint_name(m_h, (int)j)) m_active_is_revolute[i] = rdyn_chain_joint_type(m_h, (int)j) == RDYN_REVOLUTE;
    m_links.clear();
    m_joints.clear();
    for (unsigned i = 0; i < m_links_number; ++i)
    {
      LinkPtr l(new Link());
      l->m_name = m_links_name[i];
      double cog[3];
      rdyn_chain_link_parameters(m_h, (int)i, l->m_pi, &l->m_mass, cog);
      for (int k = 0; k < 3; ++k) l->m_cog(k) = cog[k];
      l->build();
      m_links.push_back(l);
    }
    for (unsigned j = 0; j < m_joints_number; ++j)
    {
      JointPtr jt(new Joint());
      jt->m_name = rdyn_chain_joint_name(m_h, (int)j);
      const int ty = rdyn_chain_joint_type(m_h, (int)j);
      jt->m_type = ty == RDYN_REVOLUTE ? Joint::REVOLUTE : (ty == RDYN_PRISMATIC ? Joint::PRISMATIC : Joint::FIXED);
      rdyn_chain_joint_constants(m_h, (int)j, jt->m_R, jt->m_t, jt->m_axis, jt->m_limits);
      jt->m_parent_link = m_links[j];
      jt->m_child_link = m_links[j + 1];
      m_links[j]->m_child_joints.push_back(jt);
      m_links[j + 1]->m_parent_joint = jt;
      m_joints.push_back(jt);
    }
    const int n = (int)m_active_joints_number;
    m_q_max.resize(n); m_q_min.resize(n); m_Dq_max.resize(n); m_DDq_max.resize(n); m_tau_max.resize(n);
    rdyn_chain_limits(m_h, m_q_max.data(), m_q_min.data(), m_Dq_max.data(), m_DDq_max.data(), m_tau_max.data());
    // device staging: 3 n inputs + the largest single-sample output (regressor n * P, frames 12 L)
    const size_t outs = std::max<size_t>((size_t)n * 10 * m_joints_number, 12 * (size_t)m_links_number) + 16;
    m_need_doubles = 3 * (size_t)n + outs;  // allocated by the first single-sample call (construction and the host-only getters need no GPU)
    m_host.assign(outs, 0.0);
    std::memset(&m_b, 0, sizeof m_b);
    m_b.n_samples = 1;
    m_b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
    m_b.device = -1;
    m_b.stream = nullptr;
  }
  double* out(size_t off) { return m_dev + 3 * (size_t)m_active_joints_number + off; }   // device address ...
  double* hout(size_t off) { return m_pin + 3 * (size_t)m_active_joints_number + off; }  // ... and host address of the same doubles
  void ensure_stage()
  {
    if (m_need_doubles <= m_dev_doubles) return;
    if (m_pin) (void)hipHostFree(m_pin);
    m_dev = m_pin = nullptr;
    m_dev_doubles = 0;
    hip(hipHostMalloc((void**)&m_pin, m_need_doubles * sizeof(double), hipHostMallocMapped));
    std::memset(m_pin, 0, m_need_doubles * sizeof(double));
    if (hipHostGetDevicePointer((void**)&m_dev, m_pin, 0) != hipSuccess)
    {
      (void)hipHostFree(m_pin);
      m_dev = m_pin = nullptr;
      throw std::runtime_error("HIP: pinned host memory is not addressable by the device");
    }
    m_dev_doubles = m_need_doubles;
  }
  void stage(const VectorXd* q, const VectorXd* dq, const VectorXd* ddq)
  {
    ensure_stage();
    const size_t n = m_active_joints_number;
    const VectorXd* src[3] = {q, dq, ddq};
    for (int k = 0; k < 3; ++k)
    {
      if (!src[k]) continue;
      if ((size_t)src[k]->rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
      std::memcpy(m_pin + k * n, src[k]->data(), n * sizeof(double));
    }
    m_b.q = m_dev;
    m_b.dq = dq ? m_dev + n : nullptr;
    m_b.ddq = ddq ? m_dev + 2 * n : nullptr;
  }
  // everything queued on the NULL stream has completed (the records are in mapped pinned memory: the kernels wrote host memory)
  void wait_done()
  {
    if (m_done_ok && !m_done_pin)
    {
      if (hipHostMalloc((void**)&m_done_pin, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&m_done_dev, m_done_pin, 0) != hipSuccess)
      {
        if (m_done_pin) (void)hipHostFree(m_done_pin);
        m_done_pin = m_done_dev = nullptr;
        m_done_ok = false;
      }
      else
        *m_done_pin = 0;
    }
    if (m_done_ok)
    {
      const uint32_t want = ++m_done_seq;
      if (hipStreamWriteValue32(nullptr, m_done_dev, want, 0) == hipSuccess)
      {
        const volatile uint32_t* const word = m_done_pin;
        for (long spins = 0; spins < 200000000L; ++spins)
          if (*word == want) return;
      }
      else
        m_done_ok = false;  // (not supported here: the plain synchronisation from now on)
    }
    hip(hipStreamSynchronize(nullptr));  // also where a failed kernel's error surfaces
  }
  void run(int status, size_t n_out)
  {
    chk(status);
    wait_done();
    std::memcpy(m_host.data(), m_pin + 3 * (size_t)m_active_joints_number, n_out * sizeof(double));
  }
  bool localIk(VectorXd& sol, const Affine3d& T_b_t, const double* weight, const VectorXd& seed, double toll, int max_iterations)
  {
    const size_t n = m_active_joints_number;
    stage(&seed, nullptr, nullptr);
    // device record after the inputs: target (12) | sol (n) | status, iterations (2 x int32 in one double)
    double* pin = m_pin + 3 * n;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r)
#ifdef RDYN_FACADE_HAS_EIGEN
        pin[c * 3 + r] = T_b_t.matrix()(r, c);
#else
        pin[c * 3 + r] = T_b_t(r, c);
#endif
    int32_t* flags = reinterpret_cast<int32_t*>(out(12 + n));
    chk(rdyn_local_ik(m_h, &m_b, out(0), weight, toll, max_iterations, out(12), flags, flags + 1));
    wait_done();
    sol.resize((int)n);
    for (size_t i = 0; i < n; ++i) sol((int)i) = pin[12 + i];
    int32_t st;
    std::memcpy(&st, pin + 12 + n, sizeof st);
    return st == 1;
  }
  const VectorOfVector6d& fill6(VectorOfVector6d& dst)
  {
    dst.resize(m_links_number);
    for (unsigned l = 0; l < m_links_number; ++l)
      for (int i = 0; i < 6; ++i) dst[l](i) = m_host[6 * l + i];
    return dst;
  }
  const VectorXd& fillv(VectorXd& dst)
  {
    dst.resize((int)m_active_joints_number);
    for (unsigned i = 0; i < m_active_joints_number; ++i) dst(i) = m_host[i];
    return dst;
  }
};

// rosdyn::createChain(urdf, base_frame, tool_frame, gravity)   primitives.h:566
inline ChainPtr createChain(const std::string& robot_description_xml, const std::string& base_frame, const std::string& tool_frame,
                            const std::array<double, 3>& gravity)
{
  return ChainPtr(new Chain(robot_description_xml, base_frame, tool_frame, gravity));
}
inline ChainPtr createChain(const ChainPtr& cpy) { return cpy->clone(); }
inline ChainPtr createChain(const Chain& cpy) { return cpy.clone(); }  // primitives.h:584

#ifdef RDYN_FACADE_HAS_URDFDOM
// rosdyn::createChain(const urdf::ModelInterface&, base_frame, tool_frame, gravity)   primitives.h:566 / primitives_impl.h:1518.
// Walks tool -> base through the parsed model exactly as Chain::init does (primitives_impl.h:600-636) and hands the flat POD
// description to the library: no XML is re-parsed.  Same exceptions and texts as the reference constructors.
namespace detail
{
inline void copy_name(char dst[64], const std::string& s)
{
  if (s.size() > 63) throw std::runtime_error("name longer than 63 characters: " + s);
  std::memset(dst, 0, 64);
  std::memcpy(dst, s.c_str(), s.size());
}
inline ChainPtr chain_from_urdfdom(const urdf::ModelInterface& model, const std::string& base_frame, const std::string& tool_frame,
                                   const double gravity[3])
{
  if (!model.getLink(base_frame)) throw std::runtime_error("Base link not found");  // primitives_impl.h:603
  urdf::LinkConstSharedPtr link = model.getLink(tool_frame);
  if (!link) throw std::runtime_error("Tool link not found");                        // primitives_impl.h:610
  std::vector<rdyn_joint_desc> joints;
  std::vector<rdyn_link_desc> links;
  auto push_link = [&](const urdf::Link& l) {
    rdyn_link_desc d;
    std::memset(&d, 0, sizeof d);
    copy_name(d.name, l.name);
    d.com_quat[3] = 1.0;
    if (l.inertial)
    {
      d.has_inertial = 1;
      d.mass = l.inertial->mass;
      d.com_xyz[0] = l.inertial->origin.position.x; d.com_xyz[1] = l.inertial->origin.position.y; d.com_xyz[2] = l.inertial->origin.position.z;
      d.com_quat[0] = l.inertial->origin.rotation.x; d.com_quat[1] = l.inertial->origin.rotation.y;
      d.com_quat[2] = l.inertial->origin.rotation.z; d.com_quat[3] = l.inertial->origin.rotation.w;
      d.ixx = l.inertial->ixx; d.ixy = l.inertial->ixy; d.ixz = l.inertial->ixz;
      d.iyy = l.inertial->iyy; d.iyz = l.inertial->iyz; d.izz = l.inertial->izz;
    }
    links.push_back(d);
  };
  // tool -> base, front-inserting (primitives_impl.h:615-626)
  while (true)
  {
    push_link(*link);
    if (link->name == base_frame) break;
    const urdf::JointSharedPtr j = link->parent_joint;
    const urdf::LinkConstSharedPtr parent = link->getParent();
    if (!j || !parent) throw std::runtime_error("Tool link is not a descendant of the base link");
    rdyn_joint_desc d;
    std::memset(&d, 0, sizeof d);
    copy_name(d.name, j->name);
    d.urdf_type = (int32_t)j->type;  // urdf::Joint::{UNKNOWN, REVOLUTE, CONTINUOUS, PRISMATIC, FLOATING, PLANAR, FIXED} = 0..6 = rdyn_urdf_joint_type
    d.origin_xyz[0] = j->parent_to_joint_origin_transform.position.x;
    d.origin_xyz[1] = j->parent_to_joint_origin_transform.position.y;
    d.origin_xyz[2] = j->parent_to_joint_origin_transform.position.z;
    d.origin_quat[0] = j->parent_to_joint_origin_transform.rotation.x;
    d.origin_quat[1] = j->parent_to_joint_origin_transform.rotation.y;
    d.origin_quat[2] = j->parent_to_joint_origin_transform.rotation.z;
    d.origin_quat[3] = j->parent_to_joint_origin_transform.rotation.w;
    d.axis[0] = j->axis.x; d.axis[1] = j->axis.y; d.axis[2] = j->axis.z;
    if (j->limits)
    {
      d.has_limits = 1;
      d.lower = j->limits->lower; d.upper = j->limits->upper; d.velocity = j->limits->velocity; d.effort = j->limits->effort;
    }
    joints.push_back(d);
    link = parent;
  }
  std::reverse(joints.begin(), joints.end());
  std::reverse(links.begin(), links.end());
  rdyn_chain_desc desc;
  desc.n_joints = (int32_t)joints.size();
  desc.joints = joints.data();
  desc.links = links.data();
  for (int i = 0; i < 3; ++i) desc.gravity[i] = gravity[i];
  return ChainPtr(new Chain(desc));
}
}  // namespace detail
inline ChainPtr createChain(const urdf::ModelInterface& urdf_model_interface, const std::string& base_frame, const std::string& tool_frame,
                            const std::array<double, 3>& gravity)
{
  return detail::chain_from_urdfdom(urdf_model_interface, base_frame, tool_frame, gravity.data());
}
#ifdef RDYN_FACADE_HAS_EIGEN
inline ChainPtr createChain(const urdf::ModelInterface& urdf_model_interface, const std::string& base_frame, const std::string& tool_frame,
                            const Eigen::Vector3d& gravity)
{
  const double g[3] = {gravity(0), gravity(1), gravity(2)};
  return detail::chain_from_urdfdom(urdf_model_interface, base_frame, tool_frame, g);
}
#endif
#endif  // RDYN_FACADE_HAS_URDFDOM

namespace detail
{
// one-sample staging for the small free-standing calls below (components, frame distance): pinned host + device buffer
struct SmallStage
{
  double* dev = nullptr;
  double* pin = nullptr;
  size_t cap = 0;
  ~SmallStage()
  {
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
  }
  void reserve(size_t doubles)
  {
    if (doubles <= cap) return;
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
    dev = pin = nullptr;
    cap = 0;
    if (hipMalloc((void**)&dev, doubles * sizeof(double)) != hipSuccess || hipHostMalloc((void**)&pin, doubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
      throw std::runtime_error("HIP: staging allocation failed");
    cap = doubles;
  }
  void up(size_t n) { if (hipMemcpyAsync(dev, pin, n * sizeof(double), hipMemcpyHostToDevice, nullptr) != hipSuccess) throw std::runtime_error("HIP: copy failed"); }
  void down(size_t off, size_t n)
  {
    if (hipMemcpyAsync(pin + off, dev + off, n * sizeof(double), hipMemcpyDeviceToHost, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess)
      throw std::runtime_error("HIP: copy failed");
  }
};
inline void status(int st)
{
  if (st == RDYN_ERR_INVALID_ARGUMENT) throw std::invalid_argument(rdyn_last_error());
  if (st != RDYN_OK) throw std::runtime_error(rdyn_last_error());
}
}  // namespace detail

// ---- per-joint additive components (base_component.h:59-189, friction_polynomial1.h:39-146, friction_polynomial2.h:36-154,
// ideal_spring.h:37-85).  Same class and method names; the constructors take the joint list and the constants directly instead
// of reading "<robot>/joint_names" and "<robot>/<joint>/<type>/{coefficients,constants}" from a ros::NodeHandle
// (base_component.h:87-99).  The regressor row is evaluated by the HIP kernel (rdyn_components_regressor), the torque variants are
// the reference's products of that row with the parameters.  descriptor() feeds the batched calls (Chain::getIdentificationGramBatch).
class ComponentBase
{
protected:
  std::string m_type, m_component_joint_name;
  std::vector<std::string> m_joint_names;
  unsigned int m_joints_number = 0, m_component_joint_number = 0;
  VectorXd m_torques, m_nominal_parameters;
  MatrixXd m_regressor;
  rdyn_component m_desc;
  detail::SmallStage m_stage;

  void computeRegressor(const VectorArg& q, const VectorArg& Dq)
  {
    const unsigned n = m_joints_number;
    if ((unsigned)q.rows() != n || (unsigned)Dq.rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
    const int K = rdyn_components_columns(&m_desc, 1);
    m_stage.reserve(2 * n + (size_t)n * K);
    std::memcpy(m_stage.pin, q.data(), n * sizeof(double));
    std::memcpy(m_stage.pin + n, Dq.data(), n * sizeof(double));
    m_stage.up(2 * n);
    rdyn_batch b;
    std::memset(&b, 0, sizeof b);
    b.n_samples = 1;
    b.q = m_stage.dev;
    b.dq = m_stage.dev + n;
    b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
    b.device = -1;
    const rdyn_regressor_layout cl = {(int64_t)n * K, 1, (int64_t)n};  // column-major n x K
    for (int k = 0; k < 3; ++k) m_desc.parameters[k] = k < (int)m_nominal_parameters.rows() ? m_nominal_parameters(k) : 0.0;
    detail::status(rdyn_components_regressor(&m_desc, 1, (int)n, &b, m_stage.dev + 2 * n, &cl, nullptr));
    m_stage.down(2 * n, (size_t)n * K);
    m_regressor.resize((int)n, K);
    std::memcpy(m_regressor.data(), m_stage.pin + 2 * n, sizeof(double) * n * K);
  }
  double rowTimesParameters(int first_col) const
  {
    double t = 0.0;
    for (int k = first_col; k < m_regressor.cols(); ++k) t += m_regressor((int)m_component_joint_number, k) * m_nominal_parameters(k);
    return t;
  }

public:
  ComponentBase(const std::string& joint_name, const std::vector<std::string>& joint_names, int type) : m_component_joint_name(joint_name), m_joint_names(joint_names)
  {
    m_joints_number = (unsigned)m_joint_names.size();
    for (m_component_joint_number = 0; m_component_joint_number < m_joints_number; ++m_component_joint_number)
      if (m_joint_names[m_component_joint_number] == m_component_joint_name) break;
    if (m_component_joint_number == m_joints_number)  // base_component.h:121-122
      throw std::invalid_argument("Component Joint name '" + m_component_joint_name + "' is not a elemente of the joint names");
    m_torques.resize((int)m_joints_number);
    std::memset(&m_desc, 0, sizeof m_desc);
    m_desc.type = type;
    m_desc.joint = (int32_t)m_component_joint_number;
  }
  virtual ~ComponentBase() {}
  ComponentBase(const ComponentBase&) = delete;
  ComponentBase& operator=(const ComponentBase&) = delete;
  virtual VectorXd getTorque(const VectorArg& q, const VectorArg& Dq, const VectorArg& DDq) = 0;
  virtual VectorXd getAdditiveTorque(const VectorArg& q, const VectorArg& Dq, const VectorArg& DDq) { return getTorque(q, Dq, DDq); }  // base_component.h:130
  virtual VectorXd getNonAdditiveTorque(const VectorArg&, const VectorArg&, const VectorArg&, const VectorArg&)  // base_component.h:135-141
  {
    VectorXd t((int)m_joints_number);
    for (unsigned i = 0; i < m_joints_number; ++i) t((int)i) = 0.0;
    return t;
  }
  virtual MatrixXd getRegressor(const VectorArg& q, const VectorArg& Dq, const VectorArg& DDq) = 0;
  unsigned int getParametersNumber() { return (unsigned)m_nominal_parameters.rows(); }
  VectorXd getNominalParameters() { return m_nominal_parameters; }
  const std::string& getJointName() const { return m_component_joint_name; }
  virtual bool setParameters(const VectorArg& parameters)
  {
    if (parameters.rows() != m_nominal_parameters.rows()) return false;  // ideal_spring.h:74-78
    m_nominal_parameters = parameters;
    return true;
  }
  // the C-ABI descriptor of this component with its current parameters (rdyn_components_regressor, rdyn_identification_gram)
  rdyn_component descriptor() const
  {
    rdyn_component d = m_desc;
    for (int k = 0; k < 3; ++k) d.parameters[k] = k < (int)m_nominal_parameters.rows() ? m_nominal_parameters(k) : 0.0;
    return d;
  }
};

// friction_polynomial1.h:39-146: regressor row [sign, omega], parameters {coloumb, viscous}
class FirstOrderPolynomialFriction : public ComponentBase
{
protected:
  double m_Dq_threshold, m_Dq_max;
  VectorXd frictionNonAdditive(const VectorArg& Dq, const VectorArg& additive_torque)
  {
    VectorXd tau = additive_torque;
    const int j = (int)m_component_joint_number;
    const double c0 = m_nominal_parameters(0);
    if (std::fabs(Dq(j)) < m_Dq_threshold)  // static condition, friction_polynomial1.h:109-117
    {
      if (std::fabs(additive_torque(j)) <= c0) tau(j) = 0;
      else if (additive_torque(j) > c0) tau(j) -= c0;
      else if (additive_torque(j) < -c0) tau(j) += c0;
    }
    else
      tau(j) += m_regressor(j, 0) * c0;
    return tau;
  }

public:
  FirstOrderPolynomialFriction(const std::string& joint_name, const std::vector<std::string>& joint_names, double coloumb, double viscous,
                               double min_velocity, double max_velocity, int type = RDYN_COMP_FRICTION1)
    : ComponentBase(joint_name, joint_names, type)
  {
    m_type = "friction";
    m_Dq_threshold = min_velocity < 1e-6 ? 1e-6 : min_velocity;  // friction_polynomial1.h:72-78
    m_Dq_max = max_velocity <= 0 ? 1.0e6 : max_velocity;         // friction_polynomial1.h:80-86
    m_desc.min_velocity = m_Dq_threshold;
    m_desc.max_velocity = m_Dq_max;
    m_nominal_parameters.resize(type == RDYN_COMP_FRICTION2 ? 3 : 2);
    m_nominal_parameters(0) = coloumb;
    m_nominal_parameters(1) = viscous;
    m_regressor.resize((int)m_joints_number, (int)m_nominal_parameters.rows());
  }
  VectorXd getTorque(const VectorArg& q, const VectorArg& Dq, const VectorArg&) override
  {
    computeRegressor(q, Dq);
    m_torques((int)m_component_joint_number) = rowTimesParameters(0);
    return m_torques;
  }
  VectorXd getAdditiveTorque(const VectorArg& q, const VectorArg& Dq, const VectorArg&) override  // viscous part only, :97-103
  {
    computeRegressor(q, Dq);
    m_torques((int)m_component_joint_number) = rowTimesParameters(1);
    return m_torques;
  }
  VectorXd getNonAdditiveTorque(const VectorArg&, const VectorArg& Dq, const VectorArg&, const VectorArg& additive_torque) override
  {
    return frictionNonAdditive(Dq, additive_torque);
  }
  MatrixXd getRegressor(const VectorArg& q, const VectorArg& Dq, const VectorArg&) override
  {
    computeRegressor(q, Dq);
    return m_regressor;
  }
  bool setParameters(const VectorArg& parameters) override  // the reference accepts any size here (:133-145); sizes are checked
  {
    return ComponentBase::setParameters(parameters);
  }
};

// friction_polynomial2.h:36-154: regressor row [sign, omega, omega^2 sign], parameters {coloumb, first_order_viscous, second_order_viscous}
class SecondOrderPolynomialFriction : public FirstOrderPolynomialFriction
{
public:
  SecondOrderPolynomialFriction(const std::string& joint_name, const std::vector<std::string>& joint_names, double coloumb,
                                double first_order_viscous, double second_order_viscous, double min_velocity, double max_velocity)
    : FirstOrderPolynomialFriction(joint_name, joint_names, coloumb, first_order_viscous, min_velocity, max_velocity, RDYN_COMP_FRICTION2)
  {
    m_nominal_parameters(2) = second_order_viscous;
  }
};

// ideal_spring.h:37-85: regressor row [q, 1], parameters {elasticity, offset_effort}.  The reference's getTorque reads
// q(m_joints_number) -- one past the end (ideal_spring.h:60); the component's own joint is used here.
class IdealSpring : public ComponentBase
{
public:
  IdealSpring(const std::string& joint_name, const std::vector<std::string>& joint_names, double elasticity, double offset_effort)
    : ComponentBase(joint_name, joint_names, RDYN_COMP_SPRING)
  {
    m_type = "spring";
    m_nominal_parameters.resize(2);
    m_nominal_parameters(0) = elasticity;
    m_nominal_parameters(1) = offset_effort;
    m_regressor.resize((int)m_joints_number, 2);
  }
  VectorXd getTorque(const VectorArg& q, const VectorArg& Dq, const VectorArg&) override
  {
    computeRegressor(q, Dq);
    m_torques((int)m_component_joint_number) = rowTimesParameters(0);
    return m_torques;
  }
  MatrixXd getRegressor(const VectorArg& q, const VectorArg& Dq, const VectorArg&) override
  {
    computeRegressor(q, Dq);
    return m_regressor;
  }
};

// ---- frame_distance.h:44-149: pose error between two frames, evaluated by k_frame_distance (one pair per call here; the batched
// form is rdyn_frame_distance on device records)
namespace detail
{
inline void frame_distance(const Affine3d& T_wa, const Affine3d& T_wb, int kind, Vector6d& distance, Matrix66d* jacobian)
{
  static thread_local SmallStage st;
  st.reserve(24 + 6 + 36);
  const Affine3d* T[2] = {&T_wa, &T_wb};
  for (int k = 0; k < 2; ++k)
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r)
#ifdef RDYN_FACADE_HAS_EIGEN
        st.pin[12 * k + c * 3 + r] = T[k]->matrix()(r, c);
#else
        st.pin[12 * k + c * 3 + r] = (*T[k])(r, c);
#endif
  st.up(24);
  status(rdyn_frame_distance(1, st.dev, st.dev + 12, RDYN_LAYOUT_SAMPLE_MAJOR, kind, st.dev + 24, jacobian ? st.dev + 30 : nullptr, -1, nullptr));
  st.down(24, jacobian ? 42 : 6);
  for (int i = 0; i < 6; ++i) distance(i) = st.pin[24 + i];
  if (jacobian)
    for (int c = 0; c < 6; ++c)
      for (int r = 0; r < 6; ++r) (*jacobian)(r, c) = st.pin[30 + c * 6 + r];
}
}  // namespace detail
inline void getFrameDistance(const Affine3d& T_wa, const Affine3d& T_wb, Vector6d& distance)          // frame_distance.h:44
{
  detail::frame_distance(T_wa, T_wb, RDYN_FRAME_DISTANCE_AXIS_ANGLE, distance, nullptr);
}
inline void getFrameDistanceQuat(const Affine3d& T_wa, const Affine3d& T_wb, Vector6d& distance)      // frame_distance.h:74
{
  detail::frame_distance(T_wa, T_wb, RDYN_FRAME_DISTANCE_QUAT, distance, nullptr);
}
inline void getFrameDistanceQuatJac(const Affine3d& T_wa, const Affine3d& T_wb, Vector6d& distance, Matrix66d& jacobian)  // :114
{
  detail::frame_distance(T_wa, T_wb, RDYN_FRAME_DISTANCE_QUAT_JAC, distance, &jacobian);
}

// ---- mixed-chain batch (BASELINE.json configs[4]): many (chain, batch) items, one launch per joint-count group
class MultiChainPlan
{
public:
  explicit MultiChainPlan(const std::vector<rdyn_multi_item>& items)
  {
    detail::status(rdyn_multi_plan_create(items.data(), (int)items.size(), &m_plan));
  }
  ~MultiChainPlan() { rdyn_multi_plan_destroy(m_plan); }
  MultiChainPlan(const MultiChainPlan&) = delete;
  MultiChainPlan& operator=(const MultiChainPlan&) = delete;
  void getRegressor(hipStream_t stream = nullptr) const { detail::status(rdyn_multi_plan_regressor(m_plan, stream)); }

private:
  rdyn_multi_plan* m_plan = nullptr;
};

}  // namespace rosdyn

#endif
