// rdyn_speed_test.cpp -- the reference's harness (rosdyn_core/test/rosdyn_speed_test.cpp:38-211) written against
// the C++ facade `rosdyn::Chain` of this repository (rosdyn_chain_facade.hpp), so it reads like the original:
// 10 000 trials, fresh q/Dq/DDq ~ U[-1,1] before every timed call (seeded splitmix64 instead of the unseeded
// Eigen setRandom of lines 111-114), mean microseconds per call, gravity (0, 0, -9.806) (lines 61-62).
//  part 1: one sample per call through the GPU (host -> device -> kernel -> host latency; drop-in behaviour);
//  part 2: the same 10 000 samples as ONE batched call per function on device-resident data.
// All nine timed calls of the reference (lines 109-192) plus getRegressor; the *Tool getters of lines 187-191 are called too.
// usage: rdyn_speed_test <urdf file> <base link> <tool link> [ntrial]
#include <chrono>
#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include <hip/hip_runtime_api.h>

#include "rosdyn_chain_facade.hpp"

static uint64_t g_state;
static double pm1()
{
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}
static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
  if (argc < 4)
  {
    std::fprintf(stderr, "usage: %s <urdf> <base> <tool> [ntrial]\n", argv[0]);
    return 2;
  }
  std::ifstream f(argv[1]);
  std::stringstream ss;
  ss << f.rdbuf();
  const int ntrial = argc > 4 ? std::atoi(argv[4]) : 10000;

  rosdyn::ChainPtr chain = rosdyn::createChain(ss.str(), argv[2], argv[3], {0, 0, -9.806});
  const unsigned int n_joints = chain->getActiveJointsNumber();
  rosdyn::VectorXd q(n_joints), Dq(n_joints), DDq(n_joints), DDDq(n_joints);
  auto draw = [&]() {
    for (unsigned i = 0; i < n_joints; ++i)
    {
      q(i) = pm1();
      Dq(i) = pm1();
      DDq(i) = pm1();
      DDDq(i) = pm1();
    }
  };
  if (argc > 5 && std::string(argv[5]) == "dump")
  {
#ifdef RDYN_FACADE_HAS_EIGEN
    std::printf("# types: Eigen (VectorXd / MatrixXd / Matrix<double, 6, Dynamic> / Affine3d: the signatures of primitives.h:452-547)\n");
#else
    std::printf("# types: facade stand-ins (no <Eigen/Core> on the include path)\n");
#endif
    // numerical dump of single-sample facade calls for the parity test (tests/test_facade.py):
    // q = 0.1 (i+1), Dq = -0.05 (i+1), DDq = 0.3 - 0.02 i
    for (unsigned i = 0; i < n_joints; ++i)
    {
      q(i) = 0.1 * (i + 1);
      Dq(i) = -0.05 * (i + 1);
      DDq(i) = 0.3 - 0.02 * i;
    }
    const rosdyn::VectorXd& tau = chain->getJointTorque(q, Dq, DDq);
    std::printf("tau");
    for (unsigned i = 0; i < n_joints; ++i) std::printf(" %.17g", tau(i));
    std::printf("\n");
    rosdyn::MatrixXd Y = chain->getRegressor(q, Dq, DDq);
    std::printf("Y");
    for (int c = 0; c < Y.cols(); ++c)
      for (int r = 0; r < Y.rows(); ++r) std::printf(" %.17g", Y(r, c));
    std::printf("\n");
    const rosdyn::MatrixXd& M = chain->getJointInertia(q);
    std::printf("M");
    for (int c = 0; c < M.cols(); ++c)
      for (int r = 0; r < M.rows(); ++r) std::printf(" %.17g", M(r, c));
    std::printf("\n");
    const rosdyn::Affine3d& T = chain->getTransformation(q);
    std::printf("T");
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r) std::printf(" %.17g", T(r, c));
    std::printf("\n");
    const rosdyn::Matrix6Xd& J = chain->getJacobian(q);
    std::printf("J");
    for (int c = 0; c < (int)n_joints; ++c)
      for (int r = 0; r < 6; ++r) std::printf(" %.17g", J(r, c));
    std::printf("\n");
    // the README's IK usage (rosdyn_core/README.md:78-84): T_base_tool of q, seed moved away from q
    rosdyn::Affine3d T_base_tool = T;
    rosdyn::VectorXd seed = q, sol;
    for (unsigned i = 0; i < n_joints; ++i) seed(i) += (i % 2 ? -0.15 : 0.2);
    const bool found = chain->computeLocalIk(sol, T_base_tool, seed, 1e-8, 50);
    std::printf("IK %d", found ? 1 : 0);
    for (unsigned i = 0; i < n_joints; ++i) std::printf(" %.17g", sol(i));
    std::printf("\n");
    {
      rosdyn::VectorXd DDDq(n_joints);
      for (unsigned i = 0; i < n_joints; ++i) DDDq(i) = 0.7 - 0.1 * i;
      rosdyn::VectorOfVector6d ext(chain->getLinksNumber());
      for (unsigned l = 0; l < chain->getLinksNumber(); ++l)
        for (int i = 0; i < 6; ++i) ext[l](i) = 0.5 * (l + 1) - 0.3 * i;
      const rosdyn::Vector6d& wt = chain->getWrench(q, Dq, DDq, ext).front();  // wrench at the base link: the whole chain
      std::printf("W");
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", wt(i));
      const rosdyn::Vector6d& jl = chain->getDDTwistLinearPartTool(q, DDDq);
      const rosdyn::Vector6d jlc = jl;
      const rosdyn::Vector6d& jn = chain->getDDTwistNonLinearPartTool(q, Dq, DDq);
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", jlc(i) + jn(i));
      const rosdyn::Vector6d& jt = chain->getDDTwistTool(q, Dq, DDq, DDDq);
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", jt(i));
      std::printf("\n");
    }
    {
      // getTwist (primitives.h:457): twists of every link, base first; getDTwist likewise; getNominalParameters (primitives.h:548)
      const rosdyn::VectorOfVector6d tw = chain->getTwist(q, Dq);
      std::printf("V");
      for (size_t l = 0; l < tw.size(); ++l)
        for (int i = 0; i < 6; ++i) std::printf(" %.17g", tw[l](i));
      std::printf("\n");
      const rosdyn::VectorOfVector6d dtw = chain->getDTwist(q, Dq, DDq);
      std::printf("D");
      for (size_t l = 0; l < dtw.size(); ++l)
        for (int i = 0; i < 6; ++i) std::printf(" %.17g", dtw[l](i));
      std::printf("\n");
      const rosdyn::VectorXd par = chain->getNominalParameters();
      std::printf("N");
      for (int i = 0; i < (int)par.rows(); ++i) std::printf(" %.17g", par(i));
      std::printf("\n");
    }
    const std::string mid = chain->getLinksName().at(chain->getLinksNumber() / 2);
    rosdyn::Matrix6Xd Jl = chain->getJacobianLink(q, mid);
    std::printf("L");
    for (int c = 0; c < (int)n_joints; ++c)
      for (int r = 0; r < 6; ++r) std::printf(" %.17g", Jl(r, c));
    std::printf("\n");
    // copy-assignment keeps a usable staging buffer: the assigned chain evaluates the same torque
    {
      rosdyn::Chain other(*chain);
      other = *chain;
      const rosdyn::VectorXd& t2 = other.getJointTorque(q, Dq, DDq);
      std::printf("A");
      for (unsigned i = 0; i < n_joints; ++i) std::printf(" %.17g", t2(i));
      std::printf("\n");
    }
    // getMultiplicity (primitives_impl.h:1470): number of configurations, then the first three
    {
      const std::vector<rosdyn::VectorXd> mt = chain->getMultiplicity(q);
      std::printf("P %zu", mt.size());
      for (size_t k = 0; k < mt.size() && k < 3; ++k)
        for (unsigned i = 0; i < n_joints; ++i) std::printf(" %.17g", mt[k](i));
      std::printf("\n");
    }
    // per-joint components on joint 1 (friction_polynomial1.h / friction_polynomial2.h / ideal_spring.h):
    // regressor row of the joint, getTorque, getAdditiveTorque, getNonAdditiveTorque of a unit additive torque
    {
      const std::vector<std::string>& names = chain->getActiveJointsName();
      rosdyn::FirstOrderPolynomialFriction f1(names.at(1), names, 0.4, 0.9, 1e-3, 0.08);
      rosdyn::SecondOrderPolynomialFriction f2(names.at(1), names, 0.4, 0.9, 0.25, 1e-3, 0.0);
      rosdyn::IdealSpring sp(names.at(1), names, 12.0, -0.7);
      rosdyn::ComponentBase* comps[3] = {&f1, &f2, &sp};
      rosdyn::VectorXd add((int)n_joints);
      for (unsigned i = 0; i < n_joints; ++i) add(i) = 1.0;
      std::printf("C");
      for (rosdyn::ComponentBase* cb : comps)
      {
        const rosdyn::MatrixXd R = cb->getRegressor(q, Dq, DDq);
        for (int k = 0; k < R.cols(); ++k) std::printf(" %.17g", R(1, k));
        std::printf(" %.17g %.17g %.17g %u", cb->getTorque(q, Dq, DDq)(1), cb->getAdditiveTorque(q, Dq, DDq)(1),
                    cb->getNonAdditiveTorque(q, Dq, DDq, add)(1), cb->getParametersNumber());
      }
      std::printf("\n");
    }
    // frame_distance.h: T(q) against T(seed)
    {
      const rosdyn::Affine3d Ta = chain->getTransformation(q);
      const rosdyn::Affine3d Tb = chain->getTransformation(seed);
      rosdyn::Vector6d d0, d1, d2;
      rosdyn::Matrix66d jac;
      rosdyn::getFrameDistance(Ta, Tb, d0);
      rosdyn::getFrameDistanceQuat(Ta, Tb, d1);
      rosdyn::getFrameDistanceQuatJac(Ta, Tb, d2, jac);
      std::printf("F");
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", d0(i));
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", d1(i));
      for (int i = 0; i < 6; ++i) std::printf(" %.17g", d2(i));
      for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 6; ++r) std::printf(" %.17g", jac(r, c));
      std::printf("\n");
    }
    // evaluateAll: one launch fills a record every getter answers from -- the same kernels, so the same bits as the calls above
    {
      const rosdyn::VectorXd tau1 = chain->getJointTorque(q, Dq, DDq);
      const rosdyn::VectorXd tnl1 = chain->getJointTorqueNonLinearPart(q, Dq);
      const rosdyn::MatrixXd Y1 = chain->getRegressor(q, Dq, DDq), M1 = chain->getJointInertia(q);
      const rosdyn::Matrix6Xd J1 = chain->getJacobian(q);
      const rosdyn::VectorOfAffine3d T1 = chain->getTransformations(q);
      const rosdyn::VectorOfVector6d tw1 = chain->getTwist(q, Dq), dtw1 = chain->getDTwist(q, Dq, DDq);
      chain->evaluateAll(q, Dq, DDq);
      double diff = chain->isEvaluated(q) ? 0.0 : 1.0;
      auto upd = [&](double a, double b) { diff = std::max(diff, std::fabs(a - b)); };
      const rosdyn::VectorXd& tau2 = chain->getJointTorque(q, Dq, DDq);
      const rosdyn::VectorXd& tnl2 = chain->getJointTorqueNonLinearPart(q, Dq);
      for (unsigned i = 0; i < n_joints; ++i) { upd(tau1(i), tau2(i)); upd(tnl1(i), tnl2(i)); }
      const rosdyn::MatrixXd Y2 = chain->getRegressor(q, Dq, DDq);
      const rosdyn::MatrixXd& M2 = chain->getJointInertia(q);
      for (int c = 0; c < Y1.cols(); ++c)
        for (int r = 0; r < Y1.rows(); ++r) upd(Y1(r, c), Y2(r, c));
      for (int c = 0; c < M1.cols(); ++c)
        for (int r = 0; r < M1.rows(); ++r) upd(M1(r, c), M2(r, c));
      const rosdyn::Matrix6Xd& J2 = chain->getJacobian(q);
      for (int c = 0; c < (int)n_joints; ++c)
        for (int r = 0; r < 6; ++r) upd(J1(r, c), J2(r, c));
      const rosdyn::VectorOfAffine3d& T2 = chain->getTransformations(q);
      const rosdyn::Affine3d& Tt = chain->getTransformation(q);
      for (size_t l = 0; l < T1.size(); ++l)
        for (int c = 0; c < 4; ++c)
          for (int r = 0; r < 3; ++r) upd(T1[l](r, c), T2[l](r, c));
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) upd(T1.back()(r, c), Tt(r, c));
      const rosdyn::VectorOfVector6d& tw2 = chain->getTwist(q, Dq);
      const rosdyn::VectorOfVector6d& dtw2 = chain->getDTwist(q, Dq, DDq);
      for (size_t l = 0; l < tw1.size(); ++l)
        for (int i = 0; i < 6; ++i) { upd(tw1[l](i), tw2[l](i)); upd(dtw1[l](i), dtw2[l](i)); }
      // other inputs: the getters evaluate their own function again and the record stays
      rosdyn::VectorXd q3 = q;
      const unsigned jm = n_joints > 1 ? 1 : 0;  // (the first joint of an arm often turns about the gravity axis: its torque ignores its angle)
      q3(jm) += 0.25;
      const double t3 = chain->getJointTorque(q3, Dq, DDq)(jm);
      std::printf("E %.17g %d %d\n", diff, chain->isEvaluated(q) ? 1 : 0, std::fabs(t3 - tau1(jm)) > 1e-9 ? 1 : 0);
    }
    // identification in C++: Gram of 4 096 seeded samples with the exact torques as measurements, host solve, residual of G x = c
    {
      const int N = 4096, n = (int)n_joints, P = 10 * (int)chain->getJointsNumber();
      std::vector<double> h((size_t)3 * N * n);
      g_state = 0x5EED0042ULL;
      for (double& v : h) v = pm1();
      double *d_in, *d_tau, *d_G;
      void* d_ws;
      const size_t ws = chain->getRegressorGramWorkspaceBytes();
      if (hipMalloc((void**)&d_in, h.size() * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_tau, sizeof(double) * N * n) != hipSuccess ||
          hipMalloc((void**)&d_G, sizeof(double) * (P * P + P + 1)) != hipSuccess || hipMalloc(&d_ws, ws) != hipSuccess)
        throw std::runtime_error("hipMalloc failed");
      if (hipMemcpy(d_in, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy failed");
      rdyn_batch b;
      std::memset(&b, 0, sizeof b);
      b.n_samples = N;
      b.q = d_in;
      b.dq = d_in + (size_t)N * n;
      b.ddq = d_in + (size_t)2 * N * n;
      b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
      b.device = -1;
      chain->getJointTorqueBatch(b, d_tau);
      chain->getRegressorGramBatch(b, d_tau, d_G, d_G + P * P, d_G + P * P + P, false, d_ws, ws);
      rosdyn::MatrixXd G(P, P);
      rosdyn::VectorXd c(P), x;
      if (hipMemcpy(G.data(), d_G, sizeof(double) * P * P, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(c.data(), d_G + P * P, sizeof(double) * P, hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("hipMemcpy failed");
      const int rank = rosdyn::Chain::solveNormalEquations(G, c, x);
      double res = 0.0, cmax = 0.0;
      for (int i = 0; i < P; ++i)
      {
        double s = -c(i);
        for (int j = 0; j < P; ++j) s += G(i, j) * x(j);
        res = std::max(res, std::fabs(s));
        cmax = std::max(cmax, std::fabs(c(i)));
      }
      std::printf("S %d %.17g", rank, res / cmax);
      for (int i = 0; i < P; ++i) std::printf(" %.17g", x(i));
      std::printf("\n");
      (void)hipFree(d_in); (void)hipFree(d_tau); (void)hipFree(d_G); (void)hipFree(d_ws);
    }
    return 0;
  }
  g_state = 0x5EED0001ULL;
  double t_pose = 0, t_jac = 0, t_vel = 0, t_linacc = 0, t_nonlinacc = 0, t_acc = 0, t_jerk = 0, t_torque = 0, t_inertia = 0, t_reg = 0, sink = 0;
  chain->getJointTorque(q, Dq, DDq);  // first use uploads the chain constants
  for (int idx = 0; idx < ntrial; idx++)
  {
    double t0;
    draw(); t0 = now_us(); sink += chain->getTransformation(q)(0, 3); t_pose += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getJacobian(q)(0, 0); t_jac += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getTwist(q, Dq).back()(0); t_vel += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getDTwistLinearPart(q, DDq).back()(0); t_linacc += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getDTwistNonLinearPart(q, Dq).back()(0); t_nonlinacc += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getDTwist(q, Dq, DDq).back()(0); t_acc += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getDDTwist(q, Dq, DDq, DDDq).back()(0); t_jerk += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getJointTorque(q, Dq, DDq)(0); t_torque += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getJointInertia(q)(0, 0); t_inertia += now_us() - t0;
    draw(); t0 = now_us(); sink += chain->getRegressor(q, Dq, DDq)(0, 0); t_reg += now_us() - t0;
    if (idx == 0)  // the tool-link shortcuts the reference calls untimed (lines 187-191)
      sink += chain->getTwistTool(q, Dq)(0) + chain->getDTwistLinearPartTool(q, DDq)(0) + chain->getDTwistNonLinearPartTool(q, Dq)(0) +
              chain->getDTwistTool(q, Dq, DDq)(0) + chain->getDDTwistTool(q, Dq, DDq, DDDq)(0);
  }
  std::printf("average on %d trials, ONE sample per call through the GPU (host round trip included):\n", ntrial);
  std::printf("computation time pose                                  = %9.3f [us]\n", t_pose / ntrial);
  std::printf("computation time jacobian                              = %9.3f [us]\n", t_jac / ntrial);
  std::printf("computation time velocity twists for all links         = %9.3f [us]\n", t_vel / ntrial);
  std::printf("computation time linear acceleration twists for all links  = %9.3f [us]\n", t_linacc / ntrial);
  std::printf("computation time non linear acc. twists for all links  = %9.3f [us]\n", t_nonlinacc / ntrial);
  std::printf("computation time acceleration twists for all links     = %9.3f [us]\n", t_acc / ntrial);
  std::printf("computation time jerk twists for all links             = %9.3f [us]\n", t_jerk / ntrial);
  std::printf("computation time joint torque                          = %9.3f [us]\n", t_torque / ntrial);
  std::printf("computation time joint inertia                         = %9.3f [us]\n", t_inertia / ntrial);
  std::printf("computation time regressor                             = %9.3f [us]\n", t_reg / ntrial);

  // ---- part 1b: every getter of a sample from ONE launch (evaluateAll), the reference's call pattern kept: pose, Jacobian,
  // twists, acceleration twists, joint torque, its non-linear part, joint inertia, regressor answered from the record
  {
    double t_all = 0, t_get = 0;
    chain->evaluateAll(q, Dq, DDq);  // first use allocates the record
    for (int idx = 0; idx < ntrial; idx++)
    {
      draw();
      double t0 = now_us();
      chain->evaluateAll(q, Dq, DDq);
      const double t1 = now_us();
      sink += chain->getTransformation(q)(0, 3) + chain->getJacobian(q)(0, 0) + chain->getTwist(q, Dq).back()(0) + chain->getDTwist(q, Dq, DDq).back()(0) +
              chain->getJointTorque(q, Dq, DDq)(0) + chain->getJointTorqueNonLinearPart(q, Dq)(0) + chain->getJointInertia(q)(0, 0) +
              chain->getTransformations(q).front()(0, 3);
      t_all += t1 - t0;
      t_get += now_us() - t1;
    }
    std::printf("computation time evaluateAll (pose of all links + jacobian + twists + acceleration twists + joint torque + non linear part +\n"
                "                 joint inertia + regressor: ONE kernel launch)    = %9.3f [us]   (+ %.3f us for eight getters answered from its record)\n",
                t_all / ntrial, t_get / ntrial);
    std::printf("reference, same laptop-class CPU figures (README.md:29-45): pose 0.76, + jacobian 1.07, + twists 1.26, + acceleration twists 1.84,\n"
                "                 + jerk 2.69, + joint torque 3.77, pose + jacobian + joint inertia 10.07 [us]\n");
  }

  // ---- part 2: the same number of samples as one batched call per function
  const int N = ntrial, n = (int)n_joints, L = (int)chain->getLinksNumber(), P = 10 * (int)chain->getJointsNumber();
  std::vector<double> h((size_t)3 * N * n);
  for (auto& x : h) x = pm1();
  double *d_in, *d_out;
  (void)hipMalloc((void**)&d_in, h.size() * sizeof(double));
  (void)hipMalloc((void**)&d_out, sizeof(double) * (size_t)N * (size_t)(n * P + n));
  (void)hipMemcpy(d_in, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  rdyn_batch b;
  std::memset(&b, 0, sizeof b);
  b.n_samples = N;
  b.q = d_in;
  b.dq = d_in + (size_t)N * n;
  b.ddq = d_in + (size_t)2 * N * n;
  b.layout = RDYN_LAYOUT_ELEMENT_MAJOR;
  b.device = -1;
  auto timed = [&](const char* name, auto&& call) {
    call();
    (void)hipDeviceSynchronize();
    const int reps = 20;
    double t0 = now_us();
    for (int r = 0; r < reps; ++r) call();
    (void)hipDeviceSynchronize();
    double us = (now_us() - t0) / reps;
    std::printf("batched %-46s = %9.3f [us] per call of %d samples = %9.5f [us] per sample\n", name, us, N, us / N);
  };
  rdyn_regressor_layout yl = {1, N, (int64_t)n * N};
  timed("pose (all links)", [&] { chain->getTransformationBatch(b, nullptr, d_out); });
  timed("jacobian", [&] { chain->getJacobianBatch(b, d_out); });
  timed("velocity + acceleration twists", [&] { chain->getTwistBatch(b, d_out, d_out + (size_t)6 * L * N); });
  timed("joint torque", [&] { chain->getJointTorqueBatch(b, d_out); });
  timed("joint inertia", [&] { chain->getJointInertiaBatch(b, d_out); });
  timed("joint torque + regressor", [&] { chain->getRegressorBatch(b, d_out, d_out + (size_t)n * N, yl); });
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return sink == 12345.678 ? 1 : 0;
}
