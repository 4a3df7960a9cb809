// tools/latency_probe.hip -- what ONE tiny call through the GPU costs on this stack, by completion mechanism (round 6: the facade's single-sample
// getters).  hipcc --offload-arch=gfx950 -O2 tools/latency_probe.hip -o tools/_build/latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_null() {}
// reads 18 doubles from (mapped) memory, a dependent chain of ~600 fp64 operations, writes 48 doubles + (optionally) a completion word
__global__ void k_work(const double* in, double* out, volatile uint32_t* flag, uint32_t seq, int chain)
{
  double s = 0.0;
  for (int i = 0; i < 18; ++i) s += in[i];
  for (int i = 0; i < chain; ++i) s = fma(s, 1.0000001, 1e-9);
  for (int i = 0; i < 48; ++i) out[i] = s + i;
  if (flag)
  {
    __threadfence_system();
    *flag = seq;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
  const bool spin_flag = argc > 1 && !std::strcmp(argv[1], "spin");
  if (spin_flag) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
  double *pin = nullptr, *dpin = nullptr, *dev = nullptr;
  CK(hipHostMalloc((void**)&pin, 4096, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&dpin, pin, 0));
  CK(hipMalloc((void**)&dev, 4096));
  std::memset(pin, 0, 4096);
  volatile uint32_t* hflag = (volatile uint32_t*)(pin + 256);
  uint32_t* dflag = (uint32_t*)(dpin + 256);
  const int T = 5000;
  uint32_t seq = 0;
  auto bench = [&](const char* name, auto fn) {
    for (int i = 0; i < 200; ++i) fn();
    const double t0 = now_us();
    for (int i = 0; i < T; ++i) fn();
    std::printf("%-86s %7.2f us per call\n", name, (now_us() - t0) / T);
    return 0;
  };
  std::printf("hipSetDeviceFlags(hipDeviceScheduleSpin): %s\n", spin_flag ? "yes" : "no");
  bench("null kernel + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, 0); (void)hipStreamSynchronize(nullptr); });
  bench("work kernel (device memory in / out) + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_work, dim3(1), dim3(1), 0, 0, dev, dev + 64, nullptr, 0u, 600); (void)hipStreamSynchronize(nullptr); });
  bench("work kernel (mapped host memory in / out) + hipStreamSynchronize", [&] { hipLaunchKernelGGL(k_work, dim3(1), dim3(1), 0, 0, dpin, dpin + 64, nullptr, 0u, 600); (void)hipStreamSynchronize(nullptr); });
  bench("work kernel (mapped) writes a completion word itself, host spins on it", [&] {
    ++seq;
    hipLaunchKernelGGL(k_work, dim3(1), dim3(1), 0, 0, dpin, dpin + 64, dflag, seq, 600);
    while (*hflag != seq) { }
  });
  CK(hipStreamSynchronize(nullptr));
  bench("work kernel (mapped) + hipStreamWriteValue32 to mapped memory, host spins", [&] {
    ++seq;
    hipLaunchKernelGGL(k_work, dim3(1), dim3(1), 0, 0, dpin, dpin + 64, nullptr, 0u, 600);
    (void)hipStreamWriteValue32(nullptr, (void*)dflag, seq, 0);
    while (*hflag != seq) { }
  });
  CK(hipStreamSynchronize(nullptr));
  bench("work kernel (mapped), chain of 3 000 dependent fma + completion word", [&] {
    ++seq;
    hipLaunchKernelGGL(k_work, dim3(1), dim3(1), 0, 0, dpin, dpin + 64, dflag, seq, 3000);
    while (*hflag != seq) { }
  });
  CK(hipStreamSynchronize(nullptr));
  return 0;
}
