/* examples/regressor_batch.c -- the C-ABI from plain C (C99): one batched getJointTorque + getRegressor call on device
 * buffers, the way a non-C++ host (cgo, JNI, ctypes, Fortran ...) would drive it.  Only include/rdyn.h and the HIP runtime.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/regressor_batch.c \
 *       -Lrosdyn_amd -lrdyn_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/rosdyn_amd -o regressor_batch
 *   ./regressor_batch tests/fixtures/ur10_like.urdf base_link wrist_3_link 100000
 */
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

int main(int argc, char** argv)
{
  const double gravity[3] = {0.0, 0.0, -9.806};
  rdyn_chain* chain = NULL;
  char* xml;
  int n, P, i;
  int64_t N, k;
  double *h_in, *d_q, *d_dq, *d_ddq, *d_tau, *d_Y, *h_tau, *pi, *h_Y0;
  rdyn_batch b;
  rdyn_regressor_layout yl;
  double err = 0.0;
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
  N = argc > 4 ? atoll(argv[4]) : 100000;
  /* synthetic trajectory, sample-major (AoS): x[s][j] */
  h_in = (double*)malloc(sizeof(double) * 3 * (size_t)N * n);
  for (k = 0; k < 3 * N * n; ++k) h_in[k] = (double)((k * 2654435761u) % 2000003u) / 1000001.5 - 1.0;
  hipMalloc((void**)&d_q, sizeof(double) * 3 * (size_t)N * n);
  d_dq = d_q + N * n;
  d_ddq = d_dq + N * n;
  hipMalloc((void**)&d_tau, sizeof(double) * (size_t)N * n);
  hipMalloc((void**)&d_Y, sizeof(double) * (size_t)N * n * P);
  hipMemcpy(d_q, h_in, sizeof(double) * 3 * (size_t)N * n, hipMemcpyHostToDevice);

  b.n_samples = N;
  b.q = d_q;
  b.dq = d_dq;
  b.ddq = d_ddq;
  b.layout = RDYN_LAYOUT_SAMPLE_MAJOR;
  b.device = -1;   /* current device */
  b.stream = NULL; /* default stream */
  /* stacked column-major (N n) x P regressor: Y(s, j, p) at Y[s * n + j + p * N * n] */
  yl.stride_sample = n;
  yl.stride_row = 1;
  yl.stride_col = N * n;
  if (rdyn_regressor(chain, &b, d_tau, d_Y, &yl) != RDYN_OK)
  {
    fprintf(stderr, "rdyn_regressor: %s\n", rdyn_last_error());
    return 1;
  }
  hipDeviceSynchronize();

  /* check the identity Y pi = tau on the first sample */
  h_tau = (double*)malloc(sizeof(double) * n);
  pi = (double*)malloc(sizeof(double) * P);
  h_Y0 = (double*)malloc(sizeof(double) * n);
  rdyn_nominal_parameters(chain, pi);
  hipMemcpy(h_tau, d_tau, sizeof(double) * n, hipMemcpyDeviceToHost);
  for (i = 0; i < n; ++i) h_tau[i] = -h_tau[i];
  for (k = 0; k < P; ++k)
  {
    hipMemcpy(h_Y0, d_Y + k * N * n, sizeof(double) * n, hipMemcpyDeviceToHost);
    for (i = 0; i < n; ++i) h_tau[i] += h_Y0[i] * pi[k];
  }
  for (i = 0; i < n; ++i) err = err > (h_tau[i] < 0 ? -h_tau[i] : h_tau[i]) ? err : (h_tau[i] < 0 ? -h_tau[i] : h_tau[i]);
  printf("chain %s -> %s: n = %d, P = %d, %lld samples, |Y pi - tau| of sample 0 = %.3e\n", argv[2], argv[3], n, P, (long long)N, err);
  rdyn_chain_destroy(chain);
  hipFree(d_q);
  hipFree(d_tau);
  hipFree(d_Y);
  return err < 1e-9 ? 0 : 1;
}
