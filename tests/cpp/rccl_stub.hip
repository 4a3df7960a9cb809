// rccl_stub.hip -- TEST INFRASTRUCTURE ONLY (never linked into or shipped with librdyn_hip.so): a stand-in for the eight RCCL entry
// points rosdyn_amd/csrc/rdyn_multi_gpu.cpp resolves at run time, for LOGICAL ranks that all live on ONE physical GPU.
//
// Why: the in-library multi-device path (ncclCommInitAll, one host thread and one stream per device, one grouped all-reduce / all-gather,
// event ordering between the context's and the callers' streams, abort on a partial group) had never executed with more than one
// device -- the pool hands out one GPU per lease and real RCCL refuses a communicator clique with repeated ordinals.  With
// RDYN_TEST_ALIAS_DEVICES=1 the library accepts device lists like {0, 0, 0, 0} (n logical devices, each with its own stream, events and
// workspace on GPU 0) and RDYN_RCCL_PATH points its dlopen at this file's .so: tests/test_multi_gpu_alias.py then runs every
// rdyn_*_multi entry point at n_dev = 2, 4, 8.
//
// Semantics kept from RCCL: calls inside ncclGroupStart / ncclGroupEnd are only queued, the collective runs at the outermost GroupEnd;
// every rank's part is ordered on ITS stream (the data a rank contributes is whatever its stream has produced by then; the results are
// visible to work queued on its stream afterwards); an all-reduce sums the ranks in rank order (deterministic); ncclCommAbort makes the
// communicator refuse further work and lets a GroupEnd return instead of waiting for ranks that never called.
// Extras for the tests: rccl_stub_fail_at(k) makes the k-th collective CALL from now fail (ncclInternalError) without queuing anything;
// rccl_stub_group_depth() is the calling thread's open-group depth; rccl_stub_collectives() counts completed collectives.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace
{
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

struct Clique;
struct Comm
{
  Clique* clique;
  int rank;
  bool aborted;
};
struct Clique
{
  int n = 0;
  int live = 0;
  std::vector<Comm*> comms;
  std::vector<hipEvent_t> ready;  // per rank: its stream has produced its contribution
  hipEvent_t done = nullptr;      // rank 0's stream has written every rank's result
  double* scratch = nullptr;
  size_t scratch_doubles = 0;
  std::mutex mu;
};
struct Op
{
  int kind;  // 0 all-reduce, 1 all-gather
  const void* send;
  void* recv;
  size_t count;
  Comm* comm;
  hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_pending;
std::atomic<long> g_fail_at{0}, g_calls{0}, g_done{0};

struct Ptrs
{
  const double* p[64];
};
__global__ void k_sum(Ptrs src, int n, double* out, size_t count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = src.p[0][i];
  for (int r = 1; r < n; ++r) s += src.p[r][i];
  out[i] = s;
}

int run_collective(std::vector<Op>& ops)
{
  // ops: one per rank of ONE clique, same kind and count (what rdyn_multi_gpu.cpp issues); ordered here by rank
  Clique* cl = ops[0].comm->clique;
  const int n = cl->n;
  if ((int)ops.size() != n) return ncclInvalidUsage;
  std::vector<Op*> by_rank(n, nullptr);
  for (auto& o : ops)
  {
    if (o.comm->clique != cl || o.kind != ops[0].kind || o.count != ops[0].count || by_rank[o.comm->rank]) return ncclInvalidUsage;
    by_rank[o.comm->rank] = &o;
  }
  std::lock_guard<std::mutex> lk(cl->mu);
  hipStream_t s0 = by_rank[0]->stream;
  for (int r = 0; r < n; ++r)
  {
    if (hipEventRecord(cl->ready[r], by_rank[r]->stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamWaitEvent(s0, cl->ready[r], 0) != hipSuccess) return ncclUnhandledCudaError;
  }
  const size_t count = ops[0].count;
  if (ops[0].kind == 0)
  {
    if (cl->scratch_doubles < count)
    {
      // (whatever still reads the old scratch runs on streams this call has just ordered s0 behind)
      if (hipStreamSynchronize(s0) != hipSuccess) return ncclUnhandledCudaError;
      if (cl->scratch) (void)hipFree(cl->scratch);
      cl->scratch = nullptr;
      cl->scratch_doubles = 0;
      if (hipMalloc((void**)&cl->scratch, count * sizeof(double)) != hipSuccess) return ncclUnhandledCudaError;
      cl->scratch_doubles = count;
    }
    Ptrs src;
    for (int r = 0; r < n; ++r) src.p[r] = (const double*)by_rank[r]->send;
    hipLaunchKernelGGL(k_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s0, src, n, cl->scratch, count);
    if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
    for (int r = 0; r < n; ++r)
      if (hipMemcpyAsync(by_rank[r]->recv, cl->scratch, count * sizeof(double), hipMemcpyDeviceToDevice, s0) != hipSuccess) return ncclUnhandledCudaError;
  }
  else
  {
    for (int r = 0; r < n; ++r)
      for (int k = 0; k < n; ++k)
        if (hipMemcpyAsync((double*)by_rank[r]->recv + (size_t)k * count, by_rank[k]->send, count * sizeof(double), hipMemcpyDeviceToDevice, s0) != hipSuccess)
          return ncclUnhandledCudaError;
  }
  if (hipEventRecord(cl->done, s0) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 1; r < n; ++r)
    if (hipStreamWaitEvent(by_rank[r]->stream, cl->done, 0) != hipSuccess) return ncclUnhandledCudaError;
  g_done.fetch_add(1);
  return ncclSuccess;
}

int flush_pending()
{
  std::vector<Op> ops;
  ops.swap(t_pending);
  if (ops.empty()) return ncclSuccess;
  for (auto& o : ops)
    if (o.comm->aborted) return ncclInternalError;  // an aborted communicator: the group is dropped, nobody waits
  // consecutive runs of n ops of one clique form one collective each
  size_t i = 0;
  while (i < ops.size())
  {
    const int n = ops[i].comm->clique->n;
    if (i + n > ops.size()) return ncclInvalidUsage;  // (real RCCL would wait for the missing ranks)
    std::vector<Op> one(ops.begin() + i, ops.begin() + i + n);
    const int r = run_collective(one);
    if (r != ncclSuccess) return r;
    i += n;
  }
  return ncclSuccess;
}

int enqueue(const Op& op)
{
  if (!op.comm || op.comm->aborted) return ncclInvalidArgument;
  const long k = g_calls.fetch_add(1) + 1;
  const long f = g_fail_at.load();
  if (f > 0 && k == f) return ncclInternalError;
  t_pending.push_back(op);
  if (t_depth == 0) return flush_pending();
  return ncclSuccess;
}
}  // namespace

extern "C"
{
int ncclCommInitAll(void** comms, int ndev, const int* devlist)
{
  if (!comms || ndev < 1 || ndev > 64 || !devlist) return ncclInvalidArgument;
  for (int i = 1; i < ndev; ++i)
    if (devlist[i] != devlist[0]) return ncclInvalidArgument;  // logical ranks of ONE physical GPU only
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(devlist[0]) != hipSuccess) return ncclUnhandledCudaError;
  Clique* cl = new Clique();
  cl->n = cl->live = ndev;
  cl->ready.assign(ndev, nullptr);
  bool ok = hipEventCreateWithFlags(&cl->done, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < ndev && ok; ++i) ok = hipEventCreateWithFlags(&cl->ready[i], hipEventDisableTiming) == hipSuccess;
  (void)hipSetDevice(prev);
  if (!ok) return ncclUnhandledCudaError;
  for (int i = 0; i < ndev; ++i)
  {
    Comm* c = new Comm{cl, i, false};
    cl->comms.push_back(c);
    comms[i] = c;
  }
  return ncclSuccess;
}
static int release(Comm* c)
{
  if (!c) return ncclInvalidArgument;
  Clique* cl = c->clique;
  bool last;
  {
    std::lock_guard<std::mutex> lk(cl->mu);
    last = --cl->live == 0;
  }
  delete c;
  if (last)
  {
    for (auto e : cl->ready)
      if (e) (void)hipEventDestroy(e);
    if (cl->done) (void)hipEventDestroy(cl->done);
    if (cl->scratch) (void)hipFree(cl->scratch);
    delete cl;
  }
  return ncclSuccess;
}
int ncclCommDestroy(void* comm) { return release((Comm*)comm); }
int ncclCommAbort(void* comm)
{
  // the communicator refuses further work; its memory is released with the clique's last member (the library aborts all of them)
  if (!comm) return ncclInvalidArgument;
  ((Comm*)comm)->aborted = true;
  return ncclSuccess;
}
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream)
{
  if (datatype != 8 || op != 0 || !send || !recv) return ncclInvalidArgument;  // ncclFloat64, ncclSum
  return enqueue(Op{0, send, recv, count, (Comm*)comm, stream});
}
int ncclAllGather(const void* send, void* recv, size_t sendcount, int datatype, void* comm, hipStream_t stream)
{
  if (datatype != 8 || !send || !recv) return ncclInvalidArgument;
  return enqueue(Op{1, send, recv, sendcount, (Comm*)comm, stream});
}
int ncclGroupStart()
{
  ++t_depth;
  return ncclSuccess;
}
int ncclGroupEnd()
{
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  return flush_pending();
}
const char* ncclGetErrorString(int r)
{
  switch (r)
  {
  case ncclSuccess: return "no error";
  case ncclUnhandledCudaError: return "unhandled cuda error";
  case ncclSystemError: return "unhandled system error";
  case ncclInternalError: return "internal error";
  case ncclInvalidArgument: return "invalid argument";
  case ncclInvalidUsage: return "invalid usage";
  default: return "unknown result code";
  }
}
// ---- test hooks
void rccl_stub_fail_at(long k)  // the k-th collective call from now fails (0: none)
{
  g_calls.store(0);
  g_fail_at.store(k);
}
int rccl_stub_group_depth() { return t_depth; }
long rccl_stub_collectives() { return g_done.load(); }
}
