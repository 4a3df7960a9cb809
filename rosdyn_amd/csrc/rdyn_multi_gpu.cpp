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
#include <cstdlib>
#include <cstring>
#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
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
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
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
  // RDYN_RCCL_PATH: an explicit library (a site's own RCCL build; tests/cpp/rccl_stub.hip for logical ranks on one GPU)
  void* h = nullptr;
  if (const char* path = getenv("RDYN_RCCL_PATH"))
  {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h)
    {
      rdyn_set_error("RDYN_RCCL_PATH=%s cannot be loaded: %s", path, dlerror());
      return RDYN_ERR_UNSUPPORTED;
    }
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
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
  r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
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
  bool broken = false;             // a collective failed half-way: the communicators were aborted, the context only waits to be destroyed
  ~rdyn_multi_gpu()
  {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < devices.size(); ++i)
    {
      if (hipSetDevice(devices[i]) != hipSuccess) continue;
      if (i < streams.size() && streams[i]) (void)hipStreamSynchronize(streams[i]);
      if (!broken && i < comms.size() && comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comms[i]);
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

// The per-device part of a call (a dozen launches each) is ISSUED by one host thread per device: queued from one thread in sequence,
// device 7 of 8 would start ~0.5 ms after device 0, as long as the kernels themselves take.  fn(i) runs with device i current and
// returns an rdyn_status; the first failure's status and message (thread-local in the workers) come back to the caller's thread.
static int for_each_device(rdyn_multi_gpu* ctx, const std::function<int(int)>& fn)
{
  const int n_dev = (int)ctx->devices.size();
  std::vector<int> status(n_dev, RDYN_OK);
  std::vector<std::string> message(n_dev);
  auto body = [&](int i) {
    if (hipSetDevice(ctx->devices[i]) != hipSuccess)
    {
      status[i] = RDYN_ERR_NO_DEVICE;
      message[i] = "hipSetDevice failed";
      return;
    }
    status[i] = fn(i);
    if (status[i] != RDYN_OK) message[i] = rdyn_last_error();
  };
  // (a thread that cannot be started -- std::system_error -- must not take the process down through the destructors of the joinable
  // ones, nor cross the extern "C" boundary: its shard runs on the caller's thread instead)
  std::vector<std::thread> workers;
  std::vector<int> inline_shards;
  for (int i = 1; i < n_dev; ++i)
  {
    try
    {
      workers.emplace_back(body, i);
    }
    catch (...)
    {
      inline_shards.push_back(i);
    }
  }
  body(0);
  for (int i : inline_shards) body(i);
  for (auto& t : workers) t.join();
  for (int i = 0; i < n_dev; ++i)
    if (status[i] != RDYN_OK)
    {
      rdyn_set_error("%s", message[i].c_str());
      return status[i];
    }
  return RDYN_OK;
}

// a collective that failed inside a group: closing the group as it stands could wait for ever for the calls that were never made --
// the communicators are aborted FIRST, then the group is closed (with aborted communicators ncclGroupEnd returns instead of waiting;
// its result is of no interest): the calling thread's group depth is back at zero, so whatever RCCL call it makes next -- a replacement
// context's ncclCommInitAll, another context's collective -- is not captured into a group that nobody will ever close.  The context
// then refuses further work.
static int abort_collective(rdyn_multi_gpu* ctx, ncclResult_t r, const char* what, int device)
{
  // (message first: ncclGroupEnd below may overwrite RCCL's own last-error state)
  rdyn_set_error("RCCL error: %s (%s on device %d); the communicators were aborted: destroy the context", g_rccl.GetErrorString(r), what, device);
  if (g_rccl.CommAbort)
    for (auto& cm : ctx->comms)
      if (cm) (void)g_rccl.CommAbort(cm);
  (void)g_rccl.GroupEnd();
  ctx->broken = true;
  return RDYN_ERR_HIP;
}
// ncclGroupEnd itself failed: the collective may have been issued on some devices only -- same consequence
static int group_end(rdyn_multi_gpu* ctx)
{
  const ncclResult_t r = g_rccl.GroupEnd();
  if (r == kNcclSuccess) return RDYN_OK;
  rdyn_set_error("RCCL error: %s (ncclGroupEnd); the communicators were aborted: destroy the context", g_rccl.GetErrorString(r));
  if (g_rccl.CommAbort)
    for (auto& cm : ctx->comms)
      if (cm) (void)g_rccl.CommAbort(cm);
  ctx->broken = true;
  return RDYN_ERR_HIP;
}

// grows the per-device workspace; whatever is queued on the context's stream has finished with the old one first
static int ensure_workspace(rdyn_multi_gpu* ctx, int i, size_t need)
{
  if (ctx->workspace_bytes[i] >= need) return RDYN_OK;
  RDYN_HIP_TRY2(hipStreamSynchronize(ctx->streams[i]));
  if (ctx->workspaces[i]) RDYN_HIP_TRY2(hipFree(ctx->workspaces[i]));
  ctx->workspaces[i] = nullptr;
  ctx->workspace_bytes[i] = 0;
  RDYN_HIP_TRY2(hipMalloc(&ctx->workspaces[i], need));
  ctx->workspace_bytes[i] = need;
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
  // RDYN_TEST_ALIAS_DEVICES=1 (test infrastructure, tests/test_multi_gpu_alias.py): repeated ordinals are accepted -- n LOGICAL devices
  // with their own streams, events and workspaces on one physical GPU, so that the n_dev > 1 code paths (one host thread per device,
  // the grouped collective, the event ordering) run on a one-GPU machine; needs a collective library that serves such a clique
  // (RDYN_RCCL_PATH = tests/cpp/rccl_stub.hip's .so: real RCCL refuses repeated ordinals)
  const char* const alias_env = getenv("RDYN_TEST_ALIAS_DEVICES");
  const bool alias = alias_env && alias_env[0] == '1';
  for (int i = 0; i < n_devices; ++i)
  {
    bool bad = devices[i] < 0;
    for (int j = 0; j < i && !alias; ++j) bad = bad || devices[i] == devices[j];
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

int rdyn_regressor_gram_multi_accumulate(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                                         double* const* acc, int accumulate)
{
  if (!ctx || !chain || !batches || !acc)
  {
    rdyn_set_error("rdyn_regressor_gram_multi: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (ctx->broken)
  {
    rdyn_set_error("rdyn_regressor_gram_multi: a collective of this context failed; destroy it");
    return RDYN_ERR_HIP;
  }
  const int n_dev = (int)ctx->devices.size();
  const int P = 10 * chain->n_joints();
  const size_t count = (size_t)P * P + P + 2;
  // ---- everything that can be refused is checked before anything is queued
  for (int i = 0; i < n_dev; ++i)
  {
    if (!acc[i] || (batches[i].device >= 0 && batches[i].device != ctx->devices[i]) || batches[i].n_samples < 0 ||
        (batches[i].n_samples > 0 && (!batches[i].q || !batches[i].dq || !batches[i].ddq)) ||
        (batches[i].layout != RDYN_LAYOUT_SAMPLE_MAJOR && batches[i].layout != RDYN_LAYOUT_ELEMENT_MAJOR))
    {
      rdyn_set_error("rdyn_regressor_gram_multi: shard %d: null accumulator or inputs, or batch.device differs from the context's device %d", i,
                     ctx->devices[i]);
      return RDYN_ERR_INVALID_ARGUMENT;
    }
  }
  const size_t ws_gram = rdyn_regressor_gram_workspace_bytes(chain, 0);
  if (ws_gram == 0)
  {
    rdyn_set_error("rdyn_regressor_gram_multi: the normal-equation entry points do not serve this chain (at most 111 columns after the reduction)");
    return RDYN_ERR_UNSUPPORTED;
  }
  // accumulate: the shard sums go to a buffer behind the Gram workspace, are all-reduced there and added to acc[i] (the caller's
  // accumulators hold the same totals on every device before and after)
  const size_t off_tmp = (ws_gram + 255) & ~(size_t)255, need = off_tmp + (accumulate ? count * sizeof(double) : 0);
  int prev = 0;
  RDYN_HIP_TRY2(hipGetDevice(&prev));
  struct Restore
  {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  // ---- every device: normal equations of its shard, on its own stream, issued by its own host thread
  int st = for_each_device(ctx, [&](int i) -> int {
    int s = ensure_workspace(ctx, i, need);
    if (s != RDYN_OK) return s;
    // the context's stream is non-blocking: order it behind whatever the caller has queued on batches[i].stream (NULL = the device's
    // default stream) -- the inputs and acc[i] may still be in production there
    RDYN_HIP_TRY2(hipEventRecord(ctx->events[i], (hipStream_t)batches[i].stream));
    RDYN_HIP_TRY2(hipStreamWaitEvent(ctx->streams[i], ctx->events[i], 0));
    rdyn_batch b = batches[i];
    b.device = ctx->devices[i];
    b.stream = ctx->streams[i];
    double* const a = accumulate ? (double*)((char*)ctx->workspaces[i] + off_tmp) : acc[i];
    s = rdyn_regressor_gram(chain, &b, tau_meas ? tau_meas[i] : nullptr, a, a + (size_t)P * P, a + (size_t)P * P + P, 0, 0, ctx->workspaces[i], ws_gram);
    if (s != RDYN_OK) return s;
    // the shard size travels as a KERNEL ARGUMENT (a pinned host word re-used by the next asynchronous call could be overwritten
    // before this call's copy has run)
    RDYN_HIP_TRY2(rdyn_launch_set_double(a + (size_t)P * P + P + 1, (double)b.n_samples, ctx->streams[i]));
    return RDYN_OK;
  });
  if (st != RDYN_OK) return st;
  // ---- ONE all-reduce of the accumulators (in place), all devices inside one group
  RDYN_NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < n_dev; ++i)
  {
    double* const a = accumulate ? (double*)((char*)ctx->workspaces[i] + off_tmp) : acc[i];
    ncclResult_t r = g_rccl.AllReduce(a, a, count, kNcclDouble, kNcclSum, ctx->comms[i], ctx->streams[i]);
    if (r != kNcclSuccess) return abort_collective(ctx, r, "ncclAllReduce", ctx->devices[i]);
  }
  if (group_end(ctx) != RDYN_OK) return RDYN_ERR_HIP;
  if (accumulate)
    for (int i = 0; i < n_dev; ++i)
    {
      RDYN_HIP_TRY2(hipSetDevice(ctx->devices[i]));
      RDYN_HIP_TRY2(rdyn_launch_add_doubles(acc[i], (const double*)((char*)ctx->workspaces[i] + off_tmp), (int64_t)count, ctx->streams[i]));
    }
  return order_callers_behind(ctx, batches);
}

int rdyn_regressor_gram_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas,
                              double* const* acc)
{
  return rdyn_regressor_gram_multi_accumulate(ctx, chain, batches, tau_meas, acc, 0);
}

int rdyn_identification_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_component* comps, int n_comps, const rdyn_batch* batches,
                                   const double* const* tau_meas, double* const* R1, int accumulate)
{
  if (!ctx || !chain || !batches || !R1 || n_comps < 0 || (n_comps > 0 && !comps))
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: null argument");
    return RDYN_ERR_INVALID_ARGUMENT;
  }
  if (ctx->broken)
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: a collective of this context failed; destroy it");
    return RDYN_ERR_HIP;
  }
  const int n_dev = (int)ctx->devices.size();
  const size_t ws_factor = rdyn_identification_tsqr_workspace_bytes(chain, comps, n_comps);
  // n1s: width of the factor of the SWEPT chain (the reduced companion of a chain with non-input joints): that is what is gathered and
  // folded -- at most 112 columns by construction, whatever the width n1 of the chain's own factor -- and expanded once per device
  int n1s = 0, n1 = 0, expands = 0;
  if (ws_factor == 0 || rdyn_internal_tsqr_widths(chain, comps, n_comps, &n1s, &n1, &expands) != RDYN_OK || n1s > rdyn_tsqr_wide_max_cols())
  {
    rdyn_set_error("rdyn_identification_tsqr_multi: the factor entry points do not serve this chain / these components");
    return RDYN_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < n_dev; ++i)
  {
    if (!R1[i] || (batches[i].device >= 0 && batches[i].device != ctx->devices[i]) || batches[i].n_samples < 0 ||
        (batches[i].n_samples > 0 && (!batches[i].q || !batches[i].dq || !batches[i].ddq)) ||
        (batches[i].layout != RDYN_LAYOUT_SAMPLE_MAJOR && batches[i].layout != RDYN_LAYOUT_ELEMENT_MAJOR))
    {
      rdyn_set_error("rdyn_identification_tsqr_multi: shard %d: null factor or inputs, or batch.device differs from the context's device %d", i,
                     ctx->devices[i]);
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
  // workspace of a device: [the factor call's | own swept factor n1s^2 | gathered factors n_dev n1s^2 | the fold's tree levels |
  //                         the folded swept factor n1s^2 | the expanded factor n1^2 (accumulating calls)]
  const size_t f_bytes = (((size_t)n1s * n1s * sizeof(double)) + 255) & ~(size_t)255, e_bytes = (((size_t)n1 * n1 * sizeof(double)) + 255) & ~(size_t)255;
  const size_t off_own = (ws_factor + 255) & ~(size_t)255, off_gather = off_own + f_bytes, off_tree = off_gather + (size_t)n_dev * f_bytes;
  const size_t off_fold = (off_tree + rdyn_tsqr_wide_workspace_doubles(n1s, n_dev) * sizeof(double) + 255) & ~(size_t)255;
  const size_t off_exp = off_fold + f_bytes, need = off_exp + (expands ? e_bytes : 0);
  int st = for_each_device(ctx, [&](int i) -> int {
    int s = ensure_workspace(ctx, i, need);
    if (s != RDYN_OK) return s;
    RDYN_HIP_TRY2(hipEventRecord(ctx->events[i], (hipStream_t)batches[i].stream));
    RDYN_HIP_TRY2(hipStreamWaitEvent(ctx->streams[i], ctx->events[i], 0));
    rdyn_batch b = batches[i];
    b.device = ctx->devices[i];
    b.stream = ctx->streams[i];
    char* const ws = (char*)ctx->workspaces[i];
    return rdyn_internal_tsqr_swept(chain, comps, n_comps, &b, tau_meas ? tau_meas[i] : nullptr, (double*)(ws + off_own), ws, ws_factor);
  });
  if (st != RDYN_OK) return st;
  // ---- ONE all-gather of the factors (n1s x n1s doubles each: the payload of the Gram all-reduce), all devices inside one group
  RDYN_NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < n_dev; ++i)
  {
    char* const ws = (char*)ctx->workspaces[i];
    ncclResult_t r = g_rccl.AllGather(ws + off_own, ws + off_gather, f_bytes / sizeof(double), kNcclDouble, ctx->comms[i], ctx->streams[i]);
    if (r != kNcclSuccess) return abort_collective(ctx, r, "ncclAllGather", ctx->devices[i]);
  }
  if (group_end(ctx) != RDYN_OK) return RDYN_ERR_HIP;
  // ---- every device folds the same stack in the same order, then expands: identical bits everywhere
  st = for_each_device(ctx, [&](int i) -> int {
    char* const ws = (char*)ctx->workspaces[i];
    double* const folded = expands ? (double*)(ws + off_fold) : R1[i];
    RDYN_HIP_TRY2(rdyn_launch_tsqr_fold_factors((const double*)(ws + off_gather), n_dev, (int64_t)(f_bytes / sizeof(double)), n1s, (double*)(ws + off_tree),
                                                folded, (!expands && accumulate) ? 1 : 0, ctx->streams[i]));
    if (expands) return rdyn_internal_tsqr_expand(chain, comps, n_comps, folded, R1[i], accumulate ? 1 : 0, (double*)(ws + off_exp), ctx->streams[i]);
    return RDYN_OK;
  });
  if (st != RDYN_OK) return st;
  return order_callers_behind(ctx, batches);
}

int rdyn_regressor_tsqr_multi(rdyn_multi_gpu* ctx, const rdyn_chain* chain, const rdyn_batch* batches, const double* const* tau_meas, double* const* R1,
                              int accumulate)
{
  return rdyn_identification_tsqr_multi(ctx, chain, nullptr, 0, batches, tau_meas, R1, accumulate);
}

}  // extern "C"
