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

#include <array>
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
#endif

namespace rosdyn
{

#ifdef RDYN_FACADE_HAS_EIGEN
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
using Matrix6Xd = Eigen::Matrix<double, 6, Eigen::Dynamic>;
using Vector6d = Eigen::Matrix<double, 6, 1>;
using Vector3d = Eigen::Vector3d;
using Affine3d = Eigen::Affine3d;
using VectorOfAffine3d = std::vector<Eigen::Affine3d, Eigen::aligned_allocator<Eigen::Affine3d>>;
using VectorOfVector6d = std::vector<Vector6d, Eigen::aligned_allocator<Vector6d>>;
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
namespace detail
{
inline void set_affine(Affine3d& T, const double* m34) { std::memcpy(T.m, m34, sizeof T.m); }
}  // namespace detail
#endif

class Chain;
using ChainPtr = std::shared_ptr<Chain>;

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

  // ---- single-sample kinematics / dynamics (primitives.h:452-463, 539-547)
  const Affine3d& getTransformation(const VectorXd& q)
  {
    stage(&q, nullptr, nullptr);
    run(rdyn_transformation(m_h, &m_b, out(0), nullptr), 12);
    detail::set_affine(m_T_bt, m_host.data());
    return m_T_bt;
  }
  const VectorOfAffine3d& getTransformations(const VectorXd& q)
  {
    stage(&q, nullptr, nullptr);
    run(rdyn_transformation(m_h, &m_b, nullptr, out(0)), 12 * m_links_number);
    m_T_bl.resize(m_links_number);
    for (unsigned int l = 0; l < m_links_number; ++l) detail::set_affine(m_T_bl[l], m_host.data() + 12 * l);
    return m_T_bl;
  }
  const Matrix6Xd& getJacobian(const VectorXd& q)
  {
    stage(&q, nullptr, nullptr);
    run(rdyn_jacobian(m_h, &m_b, out(0)), 6 * m_active_joints_number);
    m_jacobian.resize(6, (int)m_active_joints_number);
    std::memcpy(m_jacobian.data(), m_host.data(), sizeof(double) * 6 * m_active_joints_number);
    return m_jacobian;
  }
  const VectorOfVector6d& getTwist(const VectorXd& q, const VectorXd& Dq)
  {
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
    hip(hipMemcpyAsync(d_dddq, DDDq.data(), n * sizeof(double), hipMemcpyHostToDevice, nullptr));
    run(rdyn_twist_parts(m_h, &m_b, d_dddq, nullptr, nullptr, out(0)), 6 * m_links_number);
    return fill6(m_DDtwists);
  }
  // jerk split, primitives.h:476-484
  const VectorOfVector6d& getDDTwistLinearPart(const VectorXd& q, const VectorXd& DDDq)
  {
    stage(&q, nullptr, nullptr);
    if ((size_t)DDDq.rows() != m_active_joints_number) throw std::invalid_argument("Input data dimensions mismatch");
    double* d_dddq = out(6 * (size_t)m_links_number);
    hip(hipMemcpyAsync(d_dddq, DDDq.data(), m_active_joints_number * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hip(hipStreamSynchronize(nullptr));  // DDDq may be pageable caller memory
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
    std::vector<double> e(6 * (size_t)m_links_number);
    for (unsigned l = 0; l < m_links_number; ++l)
      for (int i = 0; i < 6; ++i) e[6 * l + i] = ext_wrenches_in_link_frame[l](i);
    double* d_ext = out(6 * (size_t)m_links_number);
    hip(hipMemcpyAsync(d_ext, e.data(), e.size() * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hip(hipStreamSynchronize(nullptr));  // `e` is pageable stack-owned memory
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
    std::vector<double> e(6 * (size_t)m_links_number);
    for (unsigned l = 0; l < m_links_number; ++l)
      for (int i = 0; i < 6; ++i) e[6 * l + i] = ext_wrenches_in_link_frame[l](i);
    double* d_ext = out(m_active_joints_number);
    hip(hipMemcpyAsync(d_ext, e.data(), e.size() * sizeof(double), hipMemcpyHostToDevice, nullptr));
    hip(hipStreamSynchronize(nullptr));  // `e` is pageable stack-owned memory
    run(rdyn_joint_torque_ext(m_h, &m_b, d_ext, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  const VectorXd& getJointTorque(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)
  {
    stage(&q, &Dq, &DDq);
    run(rdyn_joint_torque(m_h, &m_b, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  const VectorXd& getJointTorqueNonLinearPart(const VectorXd& q, const VectorXd& Dq)
  {
    stage(&q, &Dq, nullptr);
    run(rdyn_joint_torque_nonlinear(m_h, &m_b, out(0)), m_active_joints_number);
    return fillv(m_active_joint_torques);
  }
  MatrixXd getRegressor(const VectorXd& q, const VectorXd& Dq, const VectorXd& DDq)  // by value, primitives.h:543
  {
    if (q.rows() != Dq.rows() || Dq.rows() != DDq.rows())
      throw std::invalid_argument("Input data dimensions mismatch");  // primitives_impl.h:1299-1309
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
  VectorXd m_q_max, m_q_min, m_Dq_max, m_DDq_max, m_tau_max, m_active_joint_torques;
  Affine3d m_T_bt;
  VectorOfAffine3d m_T_bl;
  Matrix6Xd m_jacobian;
  VectorOfVector6d m_twists, m_Dtwists, m_Dtwists_linear_part, m_Dtwists_nonlinear_part, m_DDtwists, m_DDtwists_linear_part,
      m_DDtwists_nonlinear_part, m_wrenches;
  MatrixXd m_joint_inertia;
  // staging: pinned host + device buffers for ONE sample
  double* m_dev = nullptr;
  double* m_pin = nullptr;
  size_t m_dev_doubles = 0;
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
  void release()
  {
    if (m_dev) (void)hipFree(m_dev);
    if (m_pin) (void)hipHostFree(m_pin);
    m_dev = m_pin = nullptr;
    m_dev_doubles = 0;  // refresh() must allocate again (copy-assignment: release, clone, refresh)
    if (m_h) rdyn_chain_destroy(m_h);
    m_h = nullptr;
  }
  void refresh()
  {
    m_links_number = (unsigned)rdyn_chain_links_number(m_h);
    m_joints_number = (unsigned)rdyn_chain_joints_number(m_h);
    m_active_joints_number = (unsigned)rdyn_chain_active_joints_number(m_h);
    m_links_name.clear();
    m_moveable_joints_name.clear();
    m_active_joints_name.clear();
    for (unsigned i = 0; i < m_links_number; ++i) m_links_name.push_back(rdyn_chain_link_name(m_h, (int)i));
    for (int i = 0; i < rdyn_chain_moveable_joints_number(m_h); ++i) m_moveable_joints_name.push_back(rdyn_chain_moveable_joint_name(m_h, i));
    for (unsigned i = 0; i < m_active_joints_number; ++i) m_active_joints_name.push_back(rdyn_chain_active_joint_name(m_h, (int)i));
    const int n = (int)m_active_joints_number;
    m_q_max.resize(n); m_q_min.resize(n); m_Dq_max.resize(n); m_DDq_max.resize(n); m_tau_max.resize(n);
    rdyn_chain_limits(m_h, m_q_max.data(), m_q_min.data(), m_Dq_max.data(), m_DDq_max.data(), m_tau_max.data());
    // device staging: 3 n inputs + the largest single-sample output (regressor n * P, frames 12 L)
    const size_t outs = std::max<size_t>((size_t)n * 10 * m_joints_number, 12 * (size_t)m_links_number) + 16;
    const size_t need = 3 * (size_t)n + outs;
    if (need > m_dev_doubles)
    {
      if (m_dev) (void)hipFree(m_dev);
      if (m_pin) (void)hipHostFree(m_pin);
      hip(hipMalloc((void**)&m_dev, need * sizeof(double)));
      hip(hipHostMalloc((void**)&m_pin, need * sizeof(double), hipHostMallocDefault));
      m_dev_doubles = need;
    }
    m_host.assign(outs, 0.0);
    std::memset(&m_b, 0, sizeof m_b);
    m_b.n_samples = 1;
    m_b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
    m_b.device = -1;
    m_b.stream = nullptr;
  }
  double* out(size_t off) { return m_dev + 3 * (size_t)m_active_joints_number + off; }
  void stage(const VectorXd* q, const VectorXd* dq, const VectorXd* ddq)
  {
    const size_t n = m_active_joints_number;
    const VectorXd* src[3] = {q, dq, ddq};
    for (int k = 0; k < 3; ++k)
    {
      if (!src[k]) continue;
      if ((size_t)src[k]->rows() != n) throw std::invalid_argument("Input data dimensions mismatch");
      std::memcpy(m_pin + k * n, src[k]->data(), n * sizeof(double));
    }
    hip(hipMemcpyAsync(m_dev, m_pin, 3 * n * sizeof(double), hipMemcpyHostToDevice, nullptr));
    m_b.q = m_dev;
    m_b.dq = dq ? m_dev + n : nullptr;
    m_b.ddq = ddq ? m_dev + 2 * n : nullptr;
  }
  void run(int status, size_t n_out)
  {
    chk(status);
    hip(hipMemcpyAsync(m_pin + 3 * (size_t)m_active_joints_number, out(0), n_out * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    hip(hipStreamSynchronize(nullptr));
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
    hip(hipMemcpyAsync(out(0), pin, 12 * sizeof(double), hipMemcpyHostToDevice, nullptr));
    int32_t* flags = reinterpret_cast<int32_t*>(out(12 + n));
    chk(rdyn_local_ik(m_h, &m_b, out(0), weight, toll, max_iterations, out(12), flags, flags + 1));
    hip(hipMemcpyAsync(pin + 12, out(12), (n + 1) * sizeof(double), hipMemcpyDeviceToHost, nullptr));
    hip(hipStreamSynchronize(nullptr));
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

}  // namespace rosdyn

#endif
