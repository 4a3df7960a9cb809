// A CALLER of the reference's Eigen-typed public signatures on this path (primitives.h:452-547, base_component.h:124-161,
// friction_polynomial1.h:89-133, friction_polynomial2.h:95-140, ideal_spring.h:57-72), written the way code against rosdyn_core is:
// Eigen::VectorXd inputs, const Eigen::Ref<Eigen::VectorXd>& into the component classes (a writable vector binds without a copy),
// .col() / .block() on the returned Jacobian, .linear() / .translation() on the returned Affine3d.  Runs on the GPU (every getter is a
// HIP kernel behind the C-ABI) and checks the answers against identities a caller can state without the library:
//   J(q) Dq = twist of the tool link;  tau = getRegressor(q, Dq, DDq) * getNominalParameters();  R(q) R(q)' = I;
//   friction / spring: getTorque = the regressor's own row times the parameters; the closed forms of the three components.
// Built by tests/test_facade.py against tests/mock_include (no Eigen in this image) and by tools/build_with_real_eigen.sh against Eigen.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "rosdyn_chain_facade.hpp"

#ifndef RDYN_FACADE_HAS_EIGEN
#error "compile with an Eigen include path (tests/mock_include or the real headers)"
#endif

static int bad = 0;
static void expect(bool ok, const char* what)
{
  if (!ok)
  {
    std::printf("FAILED: %s\n", what);
    ++bad;
  }
}

int main(int argc, char** argv)
{
  if (argc < 4)
  {
    std::fprintf(stderr, "usage: %s <urdf> <base> <tool>\n", argv[0]);
    return 2;
  }
  std::ifstream f(argv[1]);
  std::stringstream ss;
  ss << f.rdbuf();
  rosdyn::ChainPtr chain = rosdyn::createChain(ss.str(), argv[2], argv[3], {0.0, 0.0, -9.806});
  const int n = (int)chain->getActiveJointsNumber();
  Eigen::VectorXd q(n), Dq(n), DDq(n);
  for (int i = 0; i < n; ++i)
  {
    q(i) = 0.3 * (i + 1) - 0.9;
    Dq(i) = 0.5 - 0.2 * i;
    DDq(i) = -0.4 + 0.15 * i;
  }
  // ---- Affine3d: linear() is a rotation, translation() is finite
  const Eigen::Affine3d T = chain->getTransformation(q);
  const auto R = T.linear();
  const auto p = T.translation();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += R(i, k) * R(j, k);
      expect(std::fabs(s - (i == j ? 1.0 : 0.0)) < 1e-13, "R R' = I");
    }
  expect(std::isfinite(p(0) + p(1) + p(2)), "translation finite");
  // ---- Jacobian columns: J Dq = the tool twist
  const rosdyn::Matrix6Xd J = chain->getJacobian(q);
  expect(J.rows() == 6 && J.cols() == n, "Jacobian is 6 x n");
  const rosdyn::VectorOfVector6d tw = chain->getTwist(q, Dq);
  for (int r = 0; r < 6; ++r)
  {
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += J.col(k)(r) * Dq(k);
    expect(std::fabs(s - tw.back()(r)) < 1e-12, "J Dq = tool twist");
  }
  const auto Jlin = J.block(0, 0, 3, n);
  expect(Jlin.rows() == 3 && Jlin.cols() == n && Jlin(1, 0) == J(1, 0), "block()");
  // ---- tau = Y pi
  const Eigen::VectorXd tau = chain->getJointTorque(q, Dq, DDq);
  const Eigen::MatrixXd Y = chain->getRegressor(q, Dq, DDq);
  const Eigen::VectorXd pi = chain->getNominalParameters();
  expect(Y.rows() == n && Y.cols() == pi.rows(), "regressor is n x P");
  for (int r = 0; r < n; ++r)
  {
    double s = 0.0;
    for (int k = 0; k < (int)pi.rows(); ++k) s += Y(r, k) * pi(k);
    expect(std::fabs(s - tau(r)) < 1e-11 * (1.0 + std::fabs(tau(r))), "tau = Y pi");
  }
  // ---- the component classes through const Eigen::Ref<Eigen::VectorXd>&
  const std::vector<std::string> names = chain->getActiveJointsName();
  {
    rosdyn::FirstOrderPolynomialFriction fr(names[1], names, 0.7, 1.3, 1e-3, 10.0);
    const Eigen::MatrixXd C = fr.getRegressor(q, Dq, DDq);   // VectorXd lvalues bind to Ref<VectorXd>
    expect(C.rows() == n && C.cols() == 2, "friction regressor is n x 2");
    const double sg = Dq(1) > 0 ? 1.0 : -1.0;
    expect(std::fabs(C(1, 0) - sg) < 1e-15 && std::fabs(C(1, 1) - Dq(1)) < 1e-15, "friction row = [sign, omega]");
    for (int r = 0; r < n; ++r) expect(r == 1 || (C(r, 0) == 0.0 && C(r, 1) == 0.0), "other rows zero");
    const Eigen::VectorXd t = fr.getTorque(q, Dq, DDq);
    expect(std::fabs(t(1) - (0.7 * sg + 1.3 * Dq(1))) < 1e-14, "friction torque");
    Eigen::VectorXd np(2);
    np(0) = 0.2;
    np(1) = 0.4;
    expect(fr.setParameters(np), "setParameters through Ref");
    expect(std::fabs(fr.getTorque(q, Dq, DDq)(1) - (0.2 * sg + 0.4 * Dq(1))) < 1e-14, "friction torque after setParameters");
  }
  {
    rosdyn::SecondOrderPolynomialFriction fr2(names[0], names, 0.5, 1.0, 0.05, 1e-3, 10.0);
    const Eigen::MatrixXd C = fr2.getRegressor(q, Dq, DDq);
    const double w = Dq(0), sg = w > 0 ? 1.0 : -1.0;
    expect(C.cols() == 3 && std::fabs(C(0, 2) - w * w * sg) < 1e-15, "second-order friction column");
  }
  {
    rosdyn::IdealSpring sp(names[2], names, 12.0, 0.3);
    const Eigen::MatrixXd C = sp.getRegressor(q, Dq, DDq);
    expect(C.cols() == 2 && std::fabs(C(2, 0) - q(2)) < 1e-15 && C(2, 1) == 1.0, "spring row = [q, 1]");
    expect(std::fabs(sp.getTorque(q, Dq, DDq)(2) - (12.0 * q(2) + 0.3)) < 1e-13, "spring torque");
  }
  std::printf(bad ? "%d checks failed\n" : "ok (%d joints)\n", bad ? bad : n);
  return bad ? 1 : 0;
}
