#!/bin/bash
# Host-side sanitizer run (CPU only; GPU sanitizers are not available on the pool): the URDF reader and chain ingest
# (rdyn_urdf.cpp, rdyn_chain.cpp) under ASan + UBSan over 300 fuzzed chains (tests/test_gpu_fuzz.py generator) and 400
# malformed variants of a fixture (truncations, byte flips, deleted / duplicated chunks); round 3: plus the rigid-body reduction of
# every parsed chain (as parsed, with every other input joint dropped, on a clone); round 4: plus 100 chains of up to 16 joints.
# Expected: no report.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cd "$W"
python3 - "$ROOT" <<'PY'
import os, random, sys
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
from test_gpu_fuzz import random_chain_xml, random_long_chain_xml
os.makedirs("in", exist_ok=True)
for s in range(300):
    xml, b, t, _ = random_chain_xml(5000 + s)
    open("in/f%03d.txt" % s, "w").write(b + "\n" + t + "\n" + xml)
for s in range(100):   # round 4: chains of up to 16 joints (longer than the kernels sweep: host_joints / the reduced companion carry them)
    xml, b, t, _ = random_long_chain_xml(9000 + s)
    open("in/l%03d.txt" % s, "w").write(b + "\n" + t + "\n" + xml)
rnd = random.Random(1)
base_xml = open(os.path.join(root, "tests/fixtures/mixed_joints.urdf")).read()
for s in range(400):
    x, mode = base_xml, s % 4
    i = rnd.randrange(len(x))
    if mode == 0: x = x[:i]
    elif mode == 1: x = x[:i] + rnd.choice("<>/\"'= \n&") + x[i + 1:]
    elif mode == 2: x = x[:i] + x[min(len(x), i + rnd.randrange(1, 200)):]
    else: x = x[:i] + x[i:i + rnd.randrange(1, 300)] * 2 + x[i:]
    open("in/m%03d.txt" % s, "w").write("world\ntip\n" + x)
PY
cat > harness.cpp <<'CPP'
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include "rdyn.h"
int main(int argc, char** argv)
{
  int ok = 0, bad = 0;
  for (int i = 1; i < argc; ++i)
  {
    std::ifstream f(argv[i]);
    std::string base, tool;
    std::getline(f, base);
    std::getline(f, tool);
    std::stringstream ss;
    ss << f.rdbuf();
    const double g[3] = {0, 0, -9.8};
    rdyn_chain* c = nullptr;
    if (rdyn_chain_from_urdf(ss.str().c_str(), base.c_str(), tool.c_str(), g, &c) == RDYN_OK)
    {
      ++ok;
      double pi[10 * RDYN_MAX_JOINTS], lim[5][RDYN_MAX_JOINTS];
      rdyn_nominal_parameters(c, pi);
      rdyn_chain_limits(c, lim[0], lim[1], lim[2], lim[3], lim[4]);
      for (int k = 0; k < rdyn_chain_links_number(c); ++k) (void)rdyn_chain_link_name(c, k);
      // round 3: the rigid-body reduction (rdyn_chain.cpp: build_reduced) on the chain as parsed and after dropping every other
      // input joint (non-input moving joints fold like fixed ones)
      int32_t body[RDYN_MAX_JOINTS];
      double X[RDYN_MAX_JOINTS * 100], pib[10 * RDYN_MAX_JOINTS];
      (void)rdyn_chain_reduction(c, body, X, pib);
      const int na = rdyn_chain_active_joints_number(c);
      if (na >= 2)
      {
        const char* keep[RDYN_MAX_JOINTS];
        int nk = 0;
        for (int k = 0; k < na; k += 2) keep[nk++] = rdyn_chain_active_joint_name(c, k);
        std::string names[RDYN_MAX_JOINTS];
        for (int k = 0; k < nk; ++k) names[k] = keep[k];   // the pointers die with the re-finalised chain
        for (int k = 0; k < nk; ++k) keep[k] = names[k].c_str();
        if (rdyn_chain_set_input_joints(c, keep, nk) == RDYN_OK) (void)rdyn_chain_reduction(c, body, X, pib);
      }
      rdyn_chain* d = nullptr;
      rdyn_chain_clone(c, &d);
      if (d) (void)rdyn_chain_reduction(d, body, X, pib);
      rdyn_chain_destroy(d);
      rdyn_chain_destroy(c);
    }
    else
      ++bad;
  }
  std::printf("parsed %d, rejected %d\n", ok, bad);
  return 0;
}
CPP
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ROOT/include" \
  -I"$ROOT/rosdyn_amd/csrc" harness.cpp "$ROOT/rosdyn_amd/csrc/rdyn_urdf.cpp" "$ROOT/rosdyn_amd/csrc/rdyn_chain.cpp" \
  -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o harness
ASAN_OPTIONS=detect_leaks=0 ./harness in/*.txt
rm -rf "$W"
