// tools/kbench.cpp -- torch-free A/B harness: times entry points of one or more builds of librdyn_hip.so (dlopen'ed side by
// side, interleaved rounds on the same box) and prints a checksum of the results so that variants can be compared bit for bit.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/kbench.cpp -o tools/_build/kbench \
//       -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
//   tools/_build/kbench <what> <rounds> lib1.so [lib2.so ...]        what = gram2 | gram3 | persample | stacked | element | ident | ident3 | tsqr2 | tsqr3 | itsqr2 | itsqr3 (identification R factor)
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "rdyn.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

struct Lib
{
  std::string path, label, env_key, env_val;   // "lib.so@KEY=VALUE": KEY is set around this entry's calls (probe builds)
  void* h;
  decltype(&rdyn_chain_from_urdf) chain_from_urdf;
  decltype(&rdyn_chain_active_joints_number) n_active;
  decltype(&rdyn_chain_joints_number) n_joints;
  decltype(&rdyn_regressor_gram) regressor_gram;
  decltype(&rdyn_regressor_gram_workspace_bytes) gram_ws;
  decltype(&rdyn_regressor) regressor;
  decltype(&rdyn_last_error) last_error;
  decltype(&rdyn_identification_gram) ident;
  decltype(&rdyn_identification_gram_workspace_bytes) ident_ws;
  decltype(&rdyn_regressor_tsqr) tsqr;
  decltype(&rdyn_regressor_tsqr_workspace_bytes) tsqr_ws;
  decltype(&rdyn_identification_tsqr) itsqr;
  decltype(&rdyn_identification_tsqr_workspace_bytes) itsqr_ws;
  rdyn_chain* chain = nullptr;
};

static std::string read_file(const char* p)
{
  FILE* f = std::fopen(p, "rb");
  if (!f) { std::printf("cannot read %s\n", p); std::exit(1); }
  std::string s;
  char buf[4096];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
  std::fclose(f);
  return s;
}

static uint64_t sm64(uint64_t& s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv)
{
  if (argc < 4) { std::printf("usage: kbench <gram2|gram3|persample|stacked|element|ident> <rounds> lib.so...\n"); return 2; }
  const std::string what = argv[1];
  const int rounds = std::atoi(argv[2]);
  const bool cfg3 = what == "gram3" || what == "ident3" || what == "tsqr3" || what == "itsqr3";
  // KB_URDF / KB_BASE / KB_TOOL: another chain (e.g. tests/fixtures/ur10_public.urdf base_link tool0: fixed head + two fixed tail joints)
  const char* urdf = getenv("KB_URDF") ? getenv("KB_URDF") : (cfg3 ? "tests/fixtures/panda_like.urdf" : "tests/fixtures/ur10_like.urdf");
  const char* base = getenv("KB_BASE") ? getenv("KB_BASE") : (cfg3 ? "link0" : "base_link");
  const char* tool = getenv("KB_TOOL") ? getenv("KB_TOOL") : (cfg3 ? "link7" : "wrist_3_link");
  const int64_t N = getenv("KB_N") ? std::atoll(getenv("KB_N")) : (cfg3 ? 4000000 : 1000000);
  const double g[3] = {0, 0, -9.806};
  const std::string xml = read_file(urdf);
  std::vector<Lib> libs;
  for (int i = 3; i < argc; ++i)
  {
    Lib l;
    l.label = argv[i];
    l.path = argv[i];
    const size_t at = l.path.find('@');
    if (at != std::string::npos)
    {
      const std::string kv = l.path.substr(at + 1);
      l.path = l.path.substr(0, at);
      const size_t eq = kv.find('=');
      l.env_key = kv.substr(0, eq);
      l.env_val = eq == std::string::npos ? "1" : kv.substr(eq + 1);
    }
    l.h = dlopen(l.path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { std::printf("dlopen %s: %s\n", argv[i], dlerror()); return 1; }
#define SYM(field, name) l.field = (decltype(l.field))dlsym(l.h, #name); if (!l.field) { std::printf("missing %s in %s\n", #name, argv[i]); return 1; }
    SYM(chain_from_urdf, rdyn_chain_from_urdf)
    SYM(n_active, rdyn_chain_active_joints_number)
    SYM(n_joints, rdyn_chain_joints_number)
    SYM(regressor_gram, rdyn_regressor_gram)
    SYM(gram_ws, rdyn_regressor_gram_workspace_bytes)
    SYM(regressor, rdyn_regressor)
    SYM(last_error, rdyn_last_error)
    SYM(ident, rdyn_identification_gram)
    SYM(ident_ws, rdyn_identification_gram_workspace_bytes)
    SYM(tsqr, rdyn_regressor_tsqr)
    SYM(tsqr_ws, rdyn_regressor_tsqr_workspace_bytes)
    SYM(itsqr, rdyn_identification_tsqr)
    SYM(itsqr_ws, rdyn_identification_tsqr_workspace_bytes)
    if (l.chain_from_urdf(xml.c_str(), base, tool, g, &l.chain) != RDYN_OK) { std::printf("chain: %s\n", l.last_error()); return 1; }
    libs.push_back(l);
  }
  const int n = libs[0].n_active(libs[0].chain), P = 10 * libs[0].n_joints(libs[0].chain);
  // inputs: sample-major q, dq, ddq, tau_meas
  std::vector<double> h((size_t)4 * N * n);
  uint64_t s = 0x5EED0002;
  for (auto& v : h) v = (double)(sm64(s) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
  double* d_in;
  CHECK(hipMalloc((void**)&d_in, sizeof(double) * h.size()));
  CHECK(hipMemcpy(d_in, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  rdyn_batch b;
  std::memset(&b, 0, sizeof b);
  b.n_samples = N;
  b.q = d_in;
  b.dq = d_in + (size_t)N * n;
  b.ddq = d_in + (size_t)2 * N * n;
  const double* tau_meas = d_in + (size_t)3 * N * n;
  b.layout = what == "element" ? RDYN_LAYOUT_ELEMENT_MAJOR : RDYN_LAYOUT_SAMPLE_MAJOR;
  b.device = -1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const bool gram = what.rfind("gram", 0) == 0, ident = what.rfind("ident", 0) == 0, itsqr = what.rfind("itsqr", 0) == 0, tsqr = itsqr || what.rfind("tsqr", 0) == 0;
  rdyn_component comps[7];
  const int ncomp = cfg3 ? 7 : 6;
  for (int i = 0; i < 7; ++i)
  {
    comps[i].type = getenv("KB_COMP2") ? RDYN_COMP_FRICTION2 : RDYN_COMP_FRICTION1;  // KB_COMP2: second-order friction (3 columns each)
    comps[i].joint = i;
    comps[i].min_velocity = 1e-3;
    comps[i].max_velocity = getenv("KB_COMP2") ? 10.0 : 0;
    comps[i].parameters[0] = comps[i].parameters[1] = comps[i].parameters[2] = 0.1;
  }
  const int cw = getenv("KB_COMP2") ? 3 : 2;
  const int cols = ident ? P + cw * ncomp : (itsqr ? P + cw * ncomp + 1 : (tsqr ? P + 1 : P));
  double *d_G = nullptr, *d_Y = nullptr, *d_tau = nullptr;
  void* d_ws = nullptr;
  size_t ws_bytes = 0;
  if (gram || ident || tsqr)
  {
    CHECK(hipMalloc((void**)&d_G, sizeof(double) * (cols * cols + cols + 1)));
    for (auto& l : libs)
    {
      const size_t w = ident ? l.ident_ws(l.chain, comps, ncomp) : itsqr ? l.itsqr_ws(l.chain, comps, ncomp) : (tsqr ? l.tsqr_ws(l.chain) : l.gram_ws(l.chain, 0));
      if (w > ws_bytes) ws_bytes = w;
    }
    CHECK(hipMalloc(&d_ws, ws_bytes));
  }
  // KB_BUFFERS=k (regressor modes): k output allocations kept alive, every library timed into each of them (the output-placement
  // lottery of DESIGN.md section 3: is a store schedule sensitive to WHERE the 2.88 GB land?)
  std::vector<double*> y_bufs;
  if (!(gram || ident || tsqr))
  {
    const int nbuf = getenv("KB_BUFFERS") ? std::atoi(getenv("KB_BUFFERS")) : 1;
    for (int i = 0; i < nbuf; ++i)
    {
      double* p = nullptr;
      CHECK(hipMalloc((void**)&p, sizeof(double) * (size_t)N * n * P));
      y_bufs.push_back(p);
    }
    d_Y = y_bufs[0];
    CHECK(hipMalloc((void**)&d_tau, sizeof(double) * (size_t)N * n));
  }
  rdyn_regressor_layout yl;
  if (what == "persample") yl = {(int64_t)n * P, 1, n};
  else if (what == "stacked") yl = {n, 1, N * n};
  else yl = {1, N, (int64_t)n * N};
  auto call = [&](Lib& l) -> int {
    struct Env
    {
      const Lib& l;
      explicit Env(const Lib& x) : l(x) { if (!l.env_key.empty()) setenv(l.env_key.c_str(), l.env_val.c_str(), 1); }
      ~Env() { if (!l.env_key.empty()) unsetenv(l.env_key.c_str()); }
    } env(l);
    if (itsqr) return l.itsqr(l.chain, comps, ncomp, &b, tau_meas, d_G, 0, d_ws, ws_bytes);
    if (tsqr) return l.tsqr(l.chain, &b, tau_meas, d_G, 0, d_ws, ws_bytes);
    if (ident) return l.ident(l.chain, comps, ncomp, &b, tau_meas, d_G, d_G + cols * cols, d_G + cols * cols + cols, 0, d_ws, ws_bytes);
    if (gram) return l.regressor_gram(l.chain, &b, tau_meas, d_G, d_G + P * P, d_G + P * P + P, 0, 0, d_ws, ws_bytes);
    return l.regressor(l.chain, &b, d_tau, d_Y, &yl);
  };
  std::printf("%s: n = %d, P = %d, N = %lld\n", what.c_str(), n, P, (long long)N);
  for (int r = 0; r < rounds; ++r)
   for (size_t bi = 0; bi < (y_bufs.empty() ? 1 : y_bufs.size()); ++bi)
    for (auto& l : libs)
    {
      if (!y_bufs.empty()) d_Y = y_bufs[bi];
      if (y_bufs.size() > 1) std::printf("  [buffer %zu] ", bi);
      for (int w = 0; w < 2; ++w)
        if (call(l) != RDYN_OK) { std::printf("%s: %s\n", l.path.c_str(), l.last_error()); return 1; }
      CHECK(hipDeviceSynchronize());
      const int reps = 10;
      CHECK(hipEventRecord(e0, nullptr));
      for (int k = 0; k < reps; ++k) call(l);
      CHECK(hipEventRecord(e1, nullptr));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      // checksum
      double sum = 0, sq = 0;
      if (gram || ident || tsqr)
      {
        std::vector<double> G((size_t)cols * cols + cols + 1);
        CHECK(hipMemcpy(G.data(), d_G, sizeof(double) * G.size(), hipMemcpyDeviceToHost));
        for (double v : G) { sum += v; sq += v * v; }
      }
      else
      {
        const size_t m = (size_t)N * n * P, step = 359;
        std::vector<double> Y(m);
        CHECK(hipMemcpy(Y.data(), d_Y, sizeof(double) * m, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m; i += step) { sum += Y[i]; sq += Y[i] * Y[i]; }
        CHECK(hipMemsetAsync(d_Y, 0xFF, sizeof(double) * m, nullptr));
      }
      if (getenv("KB_STAMPS") && gram)
      {
        // diagnostic builds of rdyn_duo_gram.hip (-DRDYN_DUO_STAMPS): per wave [cycles in the main loop, cycles inside barriers]
        const int NT = std::atoi(getenv("KB_STAMPS"));
        std::vector<double> st(16);
        double tot[2][2] = {{0, 0}, {0, 0}};
        for (int blk = 0; blk < 256; blk += 51)
        {
          CHECK(hipMemcpy(st.data(), (const double*)d_ws + (size_t)(256 + blk) * NT * 256, sizeof(double) * 16, hipMemcpyDeviceToHost));
          for (int w = 0; w < 8; ++w) { tot[w >= 4][0] += st[2 * w]; tot[w >= 4][1] += st[2 * w + 1]; }
        }
        std::printf("    stamps: sweeper cycles %.0f (in barriers %.0f = %.0f%%), consumer cycles %.0f (in barriers %.0f = %.0f%%)\n", tot[0][0] / 24, tot[0][1] / 24,
                    100 * tot[0][1] / tot[0][0], tot[1][0] / 24, tot[1][1] / 24, 100 * tot[1][1] / tot[1][0]);
        {
          std::vector<double> tlv(12 * 32);
          CHECK(hipMemcpy(tlv.data(), (const double*)d_ws + (size_t)600 * 4096, sizeof(double) * tlv.size(), hipMemcpyDeviceToHost));
          std::printf("    timeline of workgroup 0 (clock at entry / exit of each barrier of one trip, per wave):\n");
          for (int w = 0; w < 12; ++w)
          {
            std::printf("      w%d", w);
            for (int k = 0; k < 16; ++k) std::printf(" %7.0f", tlv[w * 32 + k] - tlv[0]);
            std::printf("\n");
          }
        }
        std::printf("    per wave of the last sampled workgroup [cycles, in barriers]:");
        for (int w = 0; w < 8; ++w) std::printf("  w%d %.0f %.0f", w, st[2 * w], st[2 * w + 1]);
        std::printf("\n");
      }
      std::printf("  %-44s %9.1f us   checksum %.17g %.17g\n", l.label.c_str(), ms * 1e3 / reps, sum, sq);
      std::fflush(stdout);
    }
  return 0;
}
