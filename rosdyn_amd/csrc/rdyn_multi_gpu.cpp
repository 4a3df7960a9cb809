// rdyn_multi_gpu.cpp -- BASELINE.json configs[3] inside the library: the trajectory batch sharded over the GPUs of one node,
// every GPU the fused regressor -> Gram of its shard (rdyn_regressor_gram: the regressor never reaches HBM), then ONE
// ncclAllReduce(P*P + P + 2 doubles, ncclDouble, ncclSum) of [G | c | bb | count] over RCCL / xGMI -- SURVEY.md section 8(e).
// Single process, one communicator per device (ncclCommInitAll), one stream per device; the collective of all devices is issued
// inside one ncclGroupStart / ncclGroupEnd.  29 KB at P = 60: latency-bound, independent of the batch size.
//
// RCCL is resolved at run time (dlopen of librccl.so.1) the first time a context is created: the library has no link-time
// dependency on it, and a process that already carries an RCCL (PyTorch bundles one under the same SONAME) shares that copy.
// Every ncclResult_t and hipError_t is checked.  No reference counterpart (rosdyn_core has no multi-device code).
//
// rdyn_identification_tsqr_multi / rdyn_regressor_tsqr_multi: the same sharding for the R factor without the normal equations
// ("alternative for TSQR"): every device the robust factor of its shard (rdyn_identification_tsqr: the
// preconditioned CholeskyQR route with its stand-by), ONE ncclAllGather of the n1 x n1 factors, and every device folds the stack of
// factors in the same fixed order (rdyn_tsqr_wide.hip) -- the result is bitwise identical on all devices.
//
// Both calls end by ordering the CALLER's stream (batches[i].stream) behind the collective with an event: work queued there
// afterwards sees the results, without a host synchronisation.
#include <dlfcn.h>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include "rdyn_chain.hpp"
#include "rdyn_kernels.h"

namespace
{

// the six entry points used, with the types of rccl.h (ncclComm_t is an opaque pointer, the enums are ints)
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { kNcclSuccess = 0, kNcclDouble = 8 /* ncclFloat64 */, kNcclSum = 0 };
struct Rccl
{
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl()
{
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return RDYN_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h)
  {
    rdyn_set_error("RCCL is not available: %s", dlerror());
    return RDYN_ERR_UNSUPPORTED;
  }
  Rccl r;
  r.handle = h;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
  {
    rdyn_set_error("RCCL is missing an entry point (ncclCommInitAll / ncclAllReduce / ncclGroup*)");
    return RDYN_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return RDYN_OK;
}

#define RDYN_NCCL_TRY(expr)                                                              \
  do                                                                                     \
  {                                                                                      \
    ncclResult_t _r = (expr);                                                            \
    if (_r != kNcclSuccess)                                                              \
    {                                                                                    \
      rdyn_set_error("RCCL error: %s (%s)", g_rccl.GetErrorString(_r), #expr);           \
      return RDYN_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)
#define RDYN_HIP_TRY2(expr)                                                              \
  do                                                                                     \
  {                                                                                      \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
    {                                                                                    \
      rdyn_set_error("HIP error: %s (%s)", hipGetErrorString(_e), #expr);                \
      return RDYN_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

}  // namespace

struct rdyn_multi_gpu
{
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<void*> workspaces;
  std::vector<size_t> workspace_bytes;
  std::vector<hipEvent_t> events;  // per device: orders the context's stream behind the caller's stream (batches[i].stream)
  std::vector<hipEvent_t> done;    // per device: orders the caller's stream behind the collective
  ~rdyn_multi_gpu()
  {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < devices.size(); ++i)
    {
      if (hipSetDevice(devices[i]) != hipSuccess) continue;
      if (i < streams.size() && streams[i]) (void)hipStreamSynchronize(streams[i]);
      if (i < comms.size() && comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comms[i]);
      if (i < workspaces.size() && workspaces[i]) (void)hipFree(workspaces[i]);
      if (i < streams.size() && streams[i]) (void)hipStreamDestroy(streams[i]);
      if (i < events.size() && events[i]) (void)hipEventDestroy(events[i]);
      if (i < done.size() && done[i]) (void)hipEventDestroy(done[i]);
    }
    (void)hipSetDevice(prev);
  }
};

// the results live on the context's (non-blocking) streams: every caller stream waits for its device's -- on the device, not the host
static int order_callers_behind(rdyn_multi_gpu* ctx, const rdyn_batch* batches)
{
  for (size_t i = 0; i < ctx->devices.size(); ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
    RDYN_HIP_TRY2(hipEventRecord(ctx->done[i], ctx->streams[i]));
    RDYN_HIP_TRY2(hipStreamWaitEvent((hipStream_t)batches[i].stream, ctx->done[i], 0));
  }
  return RDYN_OK;
}

extern "C"
{

int rdyn_multi_gpu_create(const int* devices, int n_devices, rdyn_multi_gpu** out)
{
  if (!devices || n_devices < 1 || n_devices > 64 || !out)
  {
    rdyn_set_error("rdyn_multi_gpu_create: 1..64 device ordinals and an output pointer are required");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  *out = nullptr;
  for (int i = 0; i < n_devices; ++i)
  {
    bool bad = devices[i] < 0;
    for (int j = 0; j < i; ++j) bad = bad || devices[i] == devices[j];
    if (bad)
    {
      rdyn_set_error("rdyn_multi_gpu_create: device ordinals must be distinct and non-negative");
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
  {
    rdyn_set_error("no HIP device available");
    return RDYN_ERR_NO_DEVICE;
  }
  for (int i = 0; i < n_devices; ++i)
    if (devices[i] >= count)
    {
      rdyn_set_error("rdyn_multi_gpu_create: device %d does not exist (%d devices)", devices[i], count);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  int st = load_rccl();
  if (st != RDYN_OK) return st;
  int prev = 0;
  RDYN_HIP_TRY2(hipGetDevice(&prev));
  struct Restore  // the caller's current device comes back on EVERY exit path
  {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  std::unique_ptr<rdyn_multi_gpu> ctx(new rdyn_multi_gpu());
  ctx->devices.assign(devices, devices + n_devices);
  ctx->events.assign(n_devices, nullptr);
  ctx->done.assign(n_devices, nullptr);
  ctx->comms.assign(n_devices, nullptr);
  ctx->streams.assign(n_devices, nullptr);
  ctx->workspaces.assign(n_devices, nullptr);
  ctx->workspace_bytes.assign(n_devices, 0);
  RDYN_NCCL_TRY(g_rccl.CommInitAll(ctx->comms.data(), n_devices, devices));
  for (int i = 0; i < n_devices; ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(devices[i]));
    RDYN_HIP_TRY2(hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking));
    RDYN_HIP_TRY2(hipEventCreateWithFlags(&ctx->events[i], hipEventDisableTiming));
    RDYN_HIP_TRY2(hipEventCreateWithFlags(&ctx->done[i], hipEventDisableTiming));
  }
  *out = ctx.release();
  return RDYN_OK;
}

void rdyn_multi_gpu_destroy(rdyn_multi_gpu* ctx) { delete ctx; }

int rdyn_multi_gpu_device_count(const rdyn_multi_gpu* ctx) { return ctx ? (int)ctx->devices.size() : 0; }

int rdyn_multi_gpu_synchronize(rdyn_multi_gpu* ctx)
{
  if (!ctx)
  {
    rdyn_set_error("rdyn_multi_gpu_synchronize: null context");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  int prev = 0;
  RDYN_HIP_TRY2(hipGetDevice(&prev));
  for (size_t i = 0; i < ctx->devices.size(); ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
    RDYN_HIP_TRY2(hipStreamSynchronize(ctx->streams[i]));
  }
  RDYN_HIP_TRY2(hipSetDevice(prev));
  return RDYN_OK;
}

int rdyn_regressor_gram_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                              double* const* acc)
{
  if (!ctx || !chain || !batches || !acc)
  {
    rdyn_set_error("rdyn_regressor_gram_multi: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const int n_dev = (int)ctx->devices.size();
  const int P = 10 * chain->n_joints();
  for (int i = 0; i < n_dev; ++i)
  {
    if (!acc[i] || (batches[i].device >= 0 && batches[i].device != ctx->devices[i]))
    {
      rdyn_set_error("rdyn_regressor_gram_multi: shard %d: null accumulator, or batch.device differs from the context's device %d", i, ctx->devices[i]);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  }
  int prev = 0;
  RDYN_HIP_TRY2(hipGetDevice(&prev));
  struct Restore
  {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  // ---- every device: normal equations of its shard into acc[i] = [G (P*P) | c (P) | bb (1) | count (1)], on its own stream
  for (int i = 0; i < n_dev; ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
    const size_t need = rdyn_regressor_gram_workspace_bytes(chain, 0);
    if (need == 0)
    {
      rdyn_set_error("rdyn_regressor_gram_multi: at most 111 regressor columns are supported");
      return RDYN_ERR_UNSUPPORTED;
    }
    if (ctx->workspace_bytes[i] < need)
    {
      if (ctx->workspaces[i]) RDYN_HIP_TRY2(hipFree(ctx->workspaces[i]));
      ctx->workspaces[i] = nullptr;
      ctx->workspace_bytes[i] = 0;
      RDYN_HIP_TRY2(hipMalloc(&ctx->workspaces[i], need));
      ctx->workspace_bytes[i] = need;
    }
    // the context's stream is non-blocking: order it behind whatever the caller has queued on batches[i].stream (NULL = the device's
    // default stream) -- the inputs and acc[i] may still be in production there
    RDYN_HIP_TRY2(hipEventRecord(ctx->events[i], (hipStream_t)batches[i].stream));
    RDYN_HIP_TRY2(hipStreamWaitEvent(ctx->streams[i], ctx->events[i], 0));
    rdyn_batch b = batches[i];
    b.device = ctx->devices[i];
    b.stream = ctx->streams[i];
    double* a = acc[i];
    int st = rdyn_regressor_gram(chain, &b, tau_meas ? tau_meas[i] : nullptr, a, a + (size_t)P * P, a + (size_t)P * P + P, 0, 0, ctx->workspaces[i],
                                 ctx->workspace_bytes[i]);
    if (st != RDYN_OK) return st;
    // the shard size travels as a KERNEL ARGUMENT (a pinned host word re-used by the next asynchronous call could be overwritten
    // before this call's copy has run)
    RDYN_HIP_TRY2(rdyn_launch_set_double(a + (size_t)P * P + P + 1, (double)b.n_samples, ctx->streams[i]));
  }
  // ---- ONE all-reduce of the accumulators (in place), all devices inside one group
  RDYN_NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < n_dev; ++i)
  {
    ncclResult_t r = g_rccl.AllReduce(acc[i], acc[i], (size_t)P * P + P + 2, kNcclDouble, kNcclSum, ctx->comms[i], ctx->streams[i]);
    if (r != kNcclSuccess)
    {
      (void)g_rccl.GroupEnd();
      rdyn_set_error("RCCL error: %s (ncclAllReduce on device %d)", g_rccl.GetErrorString(r), ctx->devices[i]);
      return RDYN_ERR_HIP;
    }
  }
  RDYN_NCCL_TRY(g_rccl.GroupEnd());
  return order_callers_behind(ctx, batches);
}

int rdyn_identification_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_component* comps, int n_comps, const rdyn_batch* batches,
                                   const double* const* tau_meas, double* const* R1, int accumulate)
{
  if (!ctx || !chain || !batches || !R1 || n_comps < 0 || (n_comps > 0 && !comps))
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  const int n_dev = (int)ctx->devices.size();
  const int K = n_comps > 0 ? rdyn_components_columns(comps, n_comps) : 0;
  const size_t ws_factor = rdyn_identification_tsqr_workspace_bytes(chain, comps, n_comps);
  if (K < 0 || ws_factor == 0)
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: the factor entry points do not serve this chain / these components");
    return RDYN_ERR_UNSUPPORTED;
  }
  const int n1 = 10 * chain->n_joints() + K + 1;
  if (n1 > rdyn_tsqr_wide_max_cols())
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: at most %d columns (the fold of the gathered factors)", rdyn_tsqr_wide_max_cols());
    return RDYN_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < n_dev; ++i)
  {
    if (!R1[i] || (batches[i].device >= 0 && batches[i].device != ctx->devices[i]))
    {
      rdyn_set_error("rdyn_identification_tsqr_multi: shard %d: null factor, or batch.device differs from the context's device %d", i, ctx->devices[i]);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  }
  int prev = 0;
  RDYN_HIP_TRY2(hipGetDevice(&prev));
  struct Restore
  {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  // workspace of a device: [the factor call's | own factor n1^2 | gathered factors n_dev n1^2 | the fold's tree levels]
  const size_t f_bytes = (((size_t)n1 * n1 * sizeof(double)) + 255) & ~(size_t)255;
  const size_t off_own = (ws_factor + 255) & ~(size_t)255, off_gather = off_own + f_bytes, off_tree = off_gather + (size_t)n_dev * f_bytes;
  const size_t need = off_tree + rdyn_tsqr_wide_workspace_doubles(n1, n_dev) * sizeof(double);
  for (int i = 0; i < n_dev; ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
    if (ctx->workspace_bytes[i] < need)
    {
      // (a larger workspace replaces the old one only after what is queued on the context's stream has finished with it)
      RDYN_HIP_TRY2(hipStreamSynchronize(ctx->streams[i]));
      if (ctx->workspaces[i]) RDYN_HIP_TRY2(hipFree(ctx->workspaces[i]));
      ctx->workspaces[i] = nullptr;
      ctx->workspace_bytes[i] = 0;
      RDYN_HIP_TRY2(hipMalloc(&ctx->workspaces[i], need));
      ctx->workspace_bytes[i] = need;
    }
    RDYN_HIP_TRY2(hipEventRecord(ctx->events[i], (hipStream_t)batches[i].stream));
    RDYN_HIP_TRY2(hipStreamWaitEvent(ctx->streams[i], ctx->events[i], 0));
    rdyn_batch b = batches[i];
    b.device = ctx->devices[i];
    b.stream = ctx->streams[i];
    char* const ws = (char*)ctx->workspaces[i];
    int st = rdyn_identification_tsqr(chain, comps, n_comps, &b, tau_meas ? tau_meas[i] : nullptr, (double*)(ws + off_own), 0, ws, ws_factor);
    if (st != RDYN_OK) return st;
  }
  // ---- ONE all-gather of the factors (n1 x n1 doubles each: the payload of the Gram all-reduce), all devices inside one group
  RDYN_NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < n_dev; ++i)
  {
    char* const ws = (char*)ctx->workspaces[i];
    ncclResult_t r = g_rccl.AllGather(ws + off_own, ws + off_gather, f_bytes / sizeof(double), kNcclDouble, ctx->comms[i], ctx->streams[i]);
    if (r != kNcclSuccess)
    {
      (void)g_rccl.GroupEnd();
      rdyn_set_error("RCCL error: %s (ncclAllGather on device %d)", g_rccl.GetErrorString(r), ctx->devices[i]);
      return RDYN_ERR_HIP;
    }
  }
  RDYN_NCCL_TRY(g_rccl.GroupEnd());
  // ---- every device folds the same stack in the same order: identical bits everywhere
  for (int i = 0; i < n_dev; ++i)
  {
    RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
    char* const ws = (char*)ctx->workspaces[i];
    RDYN_HIP_TRY2(rdyn_launch_tsqr_fold_factors((const double*)(ws + off_gather), n_dev, (int64_t)(f_bytes / sizeof(double)), n1, (double*)(ws + off_tree), R1[i],
                                                accumulate ? 1 : 0, ctx->streams[i]));
  }
  return order_callers_behind(ctx, batches);
}

int rdyn_regressor_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas, double* const* R1,
                              int accumulate)
{
  return rdyn_identification_tsqr_multi(ctx, chain, nullptr, 0, batches, tau_meas, R1, accumulate);
}

}  // extern "C"
