/* examples/identify.c -- the identification step from plain C (C99) on the reference's own benchmark chain in its public URDF form
 * (ur10 base_link -> tool0: a fixed joint in front, two behind; rosdyn_speed_test.cpp:44-45): measured torques of a trajectory batch ->
 *   (a) normal equations on the fp64 matrix cores   rdyn_regressor_gram  + rdyn_solve_normal_equations
 *   (b) the R factor without the normal equations   rdyn_regressor_tsqr  + rdyn_solve_r_factor
 *   (c) friction too: one FirstOrderPolynomialFriction per input joint (friction_polynomial1.h:126) stacked beside getRegressor, as the
 *       external identification step does (README.md:15): [Y | C | tau] -> rdyn_identification_tsqr + rdyn_solve_r_factor
 * -> minimum-norm inertial parameters, which must reproduce the torques, and the friction coefficients the torques were made with.
 * The regressor (N n x 90) is never stored.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/identify.c \
 *       -Lrosdyn_amd -lrdyn_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/rosdyn_amd -o identify
 *   ./identify tests/fixtures/ur10_public.urdf base_link tool0 400000
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "rdyn.h"

static char* read_file(const char* path)
{
  FILE* f = fopen(path, "rb");
  long n;
  char* s;
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  n = ftell(f);
  fseek(f, 0, SEEK_SET);
  s = (char*)malloc((size_t)n + 1);
  if (fread(s, 1, (size_t)n, f) != (size_t)n) n = 0;
  s[n] = 0;
  fclose(f);
  return s;
}

#define CHECK(call)                                                       \
  do                                                                      \
  {                                                                       \
    if ((call) != RDYN_OK)                                                \
    {                                                                     \
      fprintf(stderr, "%s: %s\n", #call, rdyn_last_error());              \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char** argv)
{
  const double gravity[3] = {0.0, 0.0, -9.806};
  rdyn_chain* chain = NULL;
  char* xml;
  int n, P, n1, i, rank_ne = 0, rank_qr = 0, bodies;
  int64_t N, k;
  double *h_in, *d_q, *d_dq, *d_ddq, *d_tau, *d_G, *d_R1, *G, *c, *R1, *x_ne, *x_qr, *h_t;
  void *ws_g, *ws_r;
  size_t nb_g, nb_r;
  rdyn_batch b;
  double e_ne = 0.0, e_qr = 0.0, tmax = 0.0, e_fr = 0.0;
  if (argc < 4)
  {
    fprintf(stderr, "usage: %s <urdf> <base link> <tool link> [samples]\n", argv[0]);
    return 2;
  }
  xml = read_file(argv[1]);
  if (!xml || rdyn_chain_from_urdf(xml, argv[2], argv[3], gravity, &chain) != RDYN_OK)
  {
    fprintf(stderr, "chain: %s\n", xml ? rdyn_last_error() : "cannot read the urdf");
    return 1;
  }
  n = rdyn_chain_active_joints_number(chain);
  P = 10 * rdyn_chain_joints_number(chain);
  n1 = P + 1;
  bodies = rdyn_chain_reduction(chain, NULL, NULL, NULL); /* > 0: links joined by fixed joints are swept as one body */
  N = argc > 4 ? atoll(argv[4]) : 400000;
  h_in = (double*)malloc(sizeof(double) * 3 * (size_t)N * n);
  for (k = 0; k < 3 * N * n; ++k) h_in[k] = (double)((k * 2654435761u) % 2000003u) / 1000001.5 - 1.0;
  hipMalloc((void**)&d_q, sizeof(double) * 3 * (size_t)N * n);
  d_dq = d_q + N * n;
  d_ddq = d_dq + N * n;
  hipMalloc((void**)&d_tau, sizeof(double) * (size_t)N * n);
  hipMemcpy(d_q, h_in, sizeof(double) * 3 * (size_t)N * n, hipMemcpyHostToDevice);
  b.n_samples = N;
  b.q = d_q;
  b.dq = d_dq;
  b.ddq = d_ddq;
  b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
  b.device = -1;
  b.stream = NULL;
  /* the "measurements": the torques of the nominal model along the trajectory */
  CHECK(rdyn_joint_torque(chain, &b, d_tau));

  /* (a) normal equations: G = A'A, c = A'tau, bb = tau'tau -- (P * P + P + 1) doubles come back */
  nb_g = rdyn_regressor_gram_workspace_bytes(chain, 0);
  hipMalloc(&ws_g, nb_g);
  hipMalloc((void**)&d_G, sizeof(double) * ((size_t)P * P + P + 1));
  CHECK(rdyn_regressor_gram(chain, &b, d_tau, d_G, d_G + (size_t)P * P, d_G + (size_t)P * P + P, 0, 0, ws_g, nb_g));
  /* (b) the R factor of [A | tau]: (P + 1)^2 doubles come back */
  nb_r = rdyn_regressor_tsqr_workspace_bytes(chain);
  hipMalloc(&ws_r, nb_r);
  hipMalloc((void**)&d_R1, sizeof(double) * (size_t)n1 * n1);
  CHECK(rdyn_regressor_tsqr(chain, &b, d_tau, d_R1, 0, ws_r, nb_r));
  hipDeviceSynchronize();
  {
    /* which stage of the factorisation vouched for the result (decided on the device, read back here) */
    rdyn_tsqr_report rep;
    CHECK(rdyn_tsqr_last_report(chain, NULL, 0, N, ws_r, -1, NULL, &rep));
    if (rep.route == 0)
      printf("R factor: Householder folds\n");
    else
      printf("R factor: preconditioned CholeskyQR, %s; %d columns deferred, growth factor %.3g, conditioning %.3g\n",
             rep.stage == 0 ? "accepted after the first round" : (rep.stage == 1 ? "accepted after the second round" : "stand-by Householder factorisation"),
             (int)rep.n_deferred, rep.gamma[rep.stage == 1 ? 1 : 0], rep.rho[rep.stage == 1 ? 1 : 0]);
  }
  G = (double*)malloc(sizeof(double) * ((size_t)P * P + P + 1));
  c = G + (size_t)P * P;
  R1 = (double*)malloc(sizeof(double) * (size_t)n1 * n1);
  hipMemcpy(G, d_G, sizeof(double) * ((size_t)P * P + P + 1), hipMemcpyDeviceToHost);
  hipMemcpy(R1, d_R1, sizeof(double) * (size_t)n1 * n1, hipMemcpyDeviceToHost);
  x_ne = (double*)malloc(sizeof(double) * P);
  x_qr = (double*)malloc(sizeof(double) * P);
  CHECK(rdyn_solve_normal_equations(G, c, P, 1e-10, x_ne, &rank_ne));
  CHECK(rdyn_solve_r_factor(R1, n1, P, P, R1 + (size_t)P * n1, 1e-9, x_qr, &rank_qr));

  /* the identified parameters (minimum norm: the regressor is structurally rank deficient) reproduce the measured torques */
  {
    /* Y x against tau for a few samples: one dense regressor call on a small prefix */
    const int64_t M = N < 4096 ? N : 4096;
    rdyn_regressor_layout yl;
    rdyn_batch bp = b;
    double* d_Y;
    double* h_Y;
    bp.n_samples = M;
    yl.stride_sample = (int64_t)n * P; /* per-sample drop-in images: Y(s, j, p) at s * n P + p * n + j */
    yl.stride_row = 1;
    yl.stride_col = n;
    hipMalloc((void**)&d_Y, sizeof(double) * (size_t)M * n * P);
    CHECK(rdyn_regressor(chain, &bp, NULL, d_Y, &yl));
    hipDeviceSynchronize();
    h_Y = (double*)malloc(sizeof(double) * (size_t)M * n * P);
    hipMemcpy(h_Y, d_Y, sizeof(double) * (size_t)M * n * P, hipMemcpyDeviceToHost);
    h_t = (double*)malloc(sizeof(double) * (size_t)M * n);
    hipMemcpy(h_t, d_tau, sizeof(double) * (size_t)M * n, hipMemcpyDeviceToHost);
    for (k = 0; k < M; ++k)
      for (i = 0; i < n; ++i)
      {
        double t_ne = 0.0, t_qr = 0.0;
        int p;
        for (p = 0; p < P; ++p)
        {
          const double y = h_Y[(size_t)k * n * P + (size_t)p * n + i];
          t_ne += y * x_ne[p];
          t_qr += y * x_qr[p];
        }
        if (fabs(h_t[k * n + i]) > tmax) tmax = fabs(h_t[k * n + i]);
        if (fabs(t_ne - h_t[k * n + i]) > e_ne) e_ne = fabs(t_ne - h_t[k * n + i]);
        if (fabs(t_qr - h_t[k * n + i]) > e_qr) e_qr = fabs(t_qr - h_t[k * n + i]);
      }
    hipFree(d_Y);
    free(h_Y);
  }
  /* (c) the same with friction: tau_meas = tau_rigid + Coulomb + viscous friction of every input joint; unknowns [inertial ; friction] */
  {
    rdyn_component comps[RDYN_MAX_SWEPT_JOINTS];
    int K, n1c, rank_c = 0, j;
    void* ws_c;
    size_t nb_c;
    double *d_R1c, *R1c, *x_c;
    for (j = 0; j < n; ++j)
    {
      comps[j].type = RDYN_COMP_FRICTION1;
      comps[j].joint = j;
      comps[j].min_velocity = 1e-3;
      comps[j].max_velocity = 10.0;
      comps[j].parameters[0] = 0.5 + 0.1 * j;  /* Coulomb */
      comps[j].parameters[1] = 1.0 + 0.2 * j;  /* viscous */
      comps[j].parameters[2] = 0.0;
    }
    K = rdyn_components_columns(comps, n);
    n1c = P + K + 1;
    CHECK(rdyn_components_regressor(comps, n, n, &b, NULL, NULL, d_tau));  /* tau += the friction torques */
    nb_c = rdyn_identification_tsqr_workspace_bytes(chain, comps, n);
    if (nb_c == 0)
    {
      fprintf(stderr, "rdyn_identification_tsqr does not serve this chain\n");
      return 1;
    }
    hipMalloc(&ws_c, nb_c);
    hipMalloc((void**)&d_R1c, sizeof(double) * (size_t)n1c * n1c);
    CHECK(rdyn_identification_tsqr(chain, comps, n, &b, d_tau, d_R1c, 0, ws_c, nb_c));
    hipDeviceSynchronize();
    R1c = (double*)malloc(sizeof(double) * (size_t)n1c * n1c);
    x_c = (double*)malloc(sizeof(double) * (size_t)(P + K));
    hipMemcpy(R1c, d_R1c, sizeof(double) * (size_t)n1c * n1c, hipMemcpyDeviceToHost);
    CHECK(rdyn_solve_r_factor(R1c, n1c, P + K, P + K, R1c + (size_t)(P + K) * n1c, 1e-9, x_c, &rank_c));
    for (j = 0; j < n; ++j)
    {
      const double ec = fabs(x_c[P + 2 * j] - comps[j].parameters[0]), ev = fabs(x_c[P + 2 * j + 1] - comps[j].parameters[1]);
      if (ec > e_fr) e_fr = ec;
      if (ev > e_fr) e_fr = ev;
    }
    printf("with %d friction components (%d columns): rank %d, max |identified - true friction coefficient| = %.3e, residual %.3e\n", n, K, rank_c, e_fr,
           fabs(R1c[(size_t)(P + K) * n1c + (P + K)]));
    hipFree(ws_c);
    hipFree(d_R1c);
    free(R1c);
    free(x_c);
  }
  printf("chain %s -> %s: n = %d, P = %d (%d rigid bodies), %lld samples\n", argv[2], argv[3], n, P, bodies, (long long)N);
  printf("normal equations: rank %d, max |Y x - tau| = %.3e;  R factor: rank %d, max |Y x - tau| = %.3e  (max |tau| = %.3e)\n", rank_ne, e_ne,
         rank_qr, e_qr, tmax);
  rdyn_chain_destroy(chain);
  hipFree(d_q);
  hipFree(d_tau);
  hipFree(d_G);
  hipFree(d_R1);
  hipFree(ws_g);
  hipFree(ws_r);
  return (e_ne < 1e-6 * tmax && e_qr < 1e-7 * tmax && rank_qr == rank_ne && e_fr < 1e-7) ? 0 : 1;
}
