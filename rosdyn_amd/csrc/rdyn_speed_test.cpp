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
    const std::string mid = chain->getLinksName().at(chain->getLinksNumber() / 2);
    rosdyn::Matrix6Xd Jl = chain->getJacobianLink(q, mid);
    std::printf("L");
    for (int c = 0; c < (int)n_joints; ++c)
      for (int r = 0; r < 6; ++r) std::printf(" %.17g", Jl(r, c));
    std::printf("\n");
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
