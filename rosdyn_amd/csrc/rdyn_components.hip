// rdyn_components.hip -- per-joint additive component regressors (friction, spring), batched.
//
// Reference (paths under /root/reference/rosdyn_core/include/rosdyn_core/):
//   FirstOrderPolynomialFriction::computeRegressor   friction_polynomial1.h:45-52   columns [sign, omega]
//   SecondOrderPolynomialFriction::computeRegressor  friction_polynomial2.h:42-58   columns [sign, omega, omega^2 sign]
//   IdealSpring::getRegressor                        ideal_spring.h:64-70           columns [q, 1]
//   ComponentBase::getTorque = regressor * nominal parameters (friction_polynomial1.h:88-94)
// These are the columns the (external) identification step appends to Chain::getRegressor.  One thread per
// sample; the component list (<= RDYN_MAX_COMPONENTS) travels in the kernel arguments (SGPRs); the output uses the
// same (sample, row, column) stride triple as the inertial regressor so the columns can be written straight
// behind it.  Pure streaming: 16 B read, n * K * 8 B written per sample.
#include <hip/hip_runtime.h>
#include "rdyn_kernels.h"

namespace
{
__global__ __launch_bounds__(256) void k_components(const RdynComponentArgs a)
{
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.n_samples) return;
  const double* __restrict__ qp = a.q + s * a.in_ss;
  const double* __restrict__ dqp = a.dq + s * a.in_ss;
  double* __restrict__ cp = a.C ? a.C + s * a.c_ss : nullptr;
  double* __restrict__ tp = a.tau ? a.tau + s * a.in_ss : nullptr;
  int k0 = 0;
  for (int i = 0; i < a.n_comps; ++i)
  {
    const RdynComponent& c = a.comps[i];
    const int cols = (c.type == RDYN_COMP_FRICTION2) ? 3 : 2;
    double row[3] = {0.0, 0.0, 0.0};
    if (c.type == RDYN_COMP_SPRING)
    {
      row[0] = qp[c.joint * a.in_sj];
      row[1] = 1.0;
    }
    else
    {
      const double v = dqp[c.joint * a.in_sj];
      const double omega = fmin(fmax(v, -c.max_velocity), c.max_velocity);
      double sg;
      if (c.type == RDYN_COMP_FRICTION1)
        sg = fmin(fmax(omega / c.min_velocity, -1.0), 1.0);
      else
        sg = (omega == 0.0) ? 0.0 : (omega > c.min_velocity ? 1.0 : (omega < -c.min_velocity ? -1.0 : omega / c.min_velocity));
      row[0] = sg;
      row[1] = omega;
      row[2] = omega * omega * sg;
    }
    if (cp)
      for (int j = 0; j < a.n_active; ++j)  // dense image: zeros outside the component's own joint row
        for (int k = 0; k < cols; ++k) cp[j * a.c_sr + (int64_t)(k0 + k) * a.c_sc] = (j == c.joint) ? row[k] : 0.0;
    if (tp)
    {
      double t = 0.0;
      for (int k = 0; k < cols; ++k) t = fma(row[k], c.parameters[k], t);
      tp[c.joint * a.in_sj] += t;
    }
    k0 += cols;
  }
}
}  // namespace

hipError_t rdyn_launch_components(const RdynComponentArgs& a, hipStream_t st)
{
  if (a.n_samples <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_components, dim3((unsigned)((a.n_samples + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}
