/*
 * components_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rosdyn_oracle.c; PARITY UNPINNED at the reference level).
 * Literal restatement of the per-joint additive component regressors of rosdyn_core
 * (paths under /root/reference/rosdyn_core/include/rosdyn_core/):
 *   FirstOrderPolynomialFriction::computeRegressor   friction_polynomial1.h:45-52   columns [sign, omega]
 *   SecondOrderPolynomialFriction::computeRegressor  friction_polynomial2.h:42-58   columns [sign, omega, omega^2 sign]
 *   IdealSpring::getRegressor                        ideal_spring.h:64-70           columns [q, 1]
 * and of the constructor rules for the constants (friction_polynomial1.h:72-86: min_velocity < 1e-6 -> 1e-6,
 * max_velocity <= 0 -> 1e6).  IdealSpring::getTorque indexes q(m_joints_number) (ideal_spring.h:60, out of range);
 * the torque used here is regressor * parameters, i.e. elasticity * q(joint) + offset_effort.
 */
#include <math.h>

enum { ORC_COMP_FRICTION1 = 0, ORC_COMP_FRICTION2 = 1, ORC_COMP_SPRING = 2 };

typedef struct
{
  int type, joint;
  double min_velocity, max_velocity;
  double parameters[3];
} orc_component;

int orc_component_columns(int type) { return type == ORC_COMP_FRICTION2 ? 3 : 2; }

/* one sample, one component: out[cols] = the component's regressor row of ITS joint */
void orc_component_row(const orc_component* c, const double* q, const double* Dq, double* out)
{
  double thr = c->min_velocity, vmax = c->max_velocity;
  if (thr < 1e-6) thr = 1e-6;        /* friction_polynomial1.h:73-78 */
  if (vmax <= 0) vmax = 1.0e6;       /* friction_polynomial1.h:81-86 */
  if (c->type == ORC_COMP_FRICTION1)
  {
    double omega = fmin(fmax(Dq[c->joint], -vmax), vmax);
    double sign_Dq = fmin(fmax(omega / thr, -1.0), 1.0);
    out[0] = sign_Dq;
    out[1] = omega;
  }
  else if (c->type == ORC_COMP_FRICTION2)
  {
    double omega = fmin(fmax(Dq[c->joint], -vmax), vmax);
    double sign_Dq = 0;
    if (omega == 0) sign_Dq = 0;
    else if (omega > thr) sign_Dq = 1.0;
    else if (omega < -thr) sign_Dq = -1.0;
    else sign_Dq = omega / thr;
    out[0] = sign_Dq;
    out[1] = omega;
    out[2] = pow(omega, 2.0) * sign_Dq;
  }
  else
  {
    out[0] = q[c->joint];
    out[1] = 1;
  }
}

/* batch: q, Dq (N, n) sample-major; C (N, n, K) with K = total columns, zero outside each component's joint row;
 * tau_add (N, n) += regressor * nominal parameters (getTorque) when non-NULL */
void orc_components_batch(const orc_component* comps, int n_comps, int n, long N, const double* q, const double* Dq, double* C, double* tau_add)
{
  int K = 0;
  for (int i = 0; i < n_comps; i++) K += orc_component_columns(comps[i].type);
  for (long s = 0; s < N; s++)
  {
    double* Cs = C + (long)s * n * K;
    for (int i = 0; i < n * K; i++) Cs[i] = 0.0;
    int k0 = 0;
    for (int i = 0; i < n_comps; i++)
    {
      double row[3];
      int cols = orc_component_columns(comps[i].type);
      orc_component_row(&comps[i], q + s * n, Dq + s * n, row);
      for (int k = 0; k < cols; k++)
      {
        Cs[comps[i].joint * K + k0 + k] = row[k];
        if (tau_add) tau_add[s * n + comps[i].joint] += row[k] * comps[i].parameters[k];
      }
      k0 += cols;
    }
  }
}
