#!/bin/bash
# tools/build_with_real_eigen.sh [eigen include dir] -- the check this image cannot run (Eigen is not installed here: the facade's Eigen-typed
# branch is compiled against the 100-line stand-in tests/mock_include/Eigen): on a machine WITH Eigen 3 and a GPU, run it unchanged.
#   1. builds librdyn_hip.so if it is missing,
#   2. compiles the reference's harness port (rdyn_speed_test.cpp), the typed-surface check and the Eigen CALLER test
#      (tests/cpp/eigen_caller.cpp: const Eigen::Ref<Eigen::VectorXd>& into the component classes, .col() / .block() on the Jacobian,
#      .linear() / .translation() on the Affine3d) with -Wall -Wextra -Werror -pedantic against the REAL headers,
#   3. runs them on the UR10-like and Panda-like fixtures (exit code 0 = every check passed).
# Eigen is looked for in: $1, $EIGEN3_INCLUDE_DIR, pkg-config eigen3, /usr/include/eigen3, /usr/local/include/eigen3.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
INC=${1:-$EIGEN3_INCLUDE_DIR}
if [ -z "$INC" ] && command -v pkg-config > /dev/null && pkg-config --exists eigen3; then INC=$(pkg-config --cflags-only-I eigen3 | sed 's/-I//; s/ .*//'); fi
for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/homebrew/include/eigen3; do [ -z "$INC" ] && [ -f $d/Eigen/Core ] && INC=$d; done
if [ -z "$INC" ] || [ ! -f "$INC/Eigen/Core" ]; then echo "Eigen 3 not found (pass its include directory as the first argument)"; exit 3; fi
if grep -q "MOCK_EIGEN_CORE" "$INC/Eigen/Core"; then echo "$INC is the test stand-in, not Eigen"; exit 3; fi
echo "Eigen: $INC ($(grep -h 'define EIGEN_\(WORLD\|MAJOR\|MINOR\)_VERSION' $INC/Eigen/src/Core/util/Macros.h | awk '{printf "%s.", $3}' | sed 's/\.$//'))"
[ -f $ROOT/rosdyn_amd/librdyn_hip.so ] || make -C $ROOT/rosdyn_amd/csrc -j"$(nproc)"
OUT=$ROOT/tools/_build/real_eigen
mkdir -p $OUT
FLAGS="-std=c++17 -O2 -Wall -Wextra -Werror -pedantic -D__HIP_PLATFORM_AMD__ -isystem /opt/rocm/include -isystem $INC -I$ROOT/rosdyn_amd/csrc"
LIBS="-L$ROOT/rosdyn_amd -lrdyn_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$ROOT/rosdyn_amd -Wl,-rpath,/opt/rocm/lib"
g++ $FLAGS $ROOT/rosdyn_amd/csrc/rdyn_speed_test.cpp -o $OUT/rdyn_speed_test_eigen $LIBS
g++ $FLAGS $ROOT/tests/cpp/eigen_caller.cpp -o $OUT/eigen_caller $LIBS
# (the typed-surface check also needs urdfdom_headers; built when <urdf_model/model.h> is found, skipped otherwise)
if echo '#include <urdf_model/model.h>' | g++ -std=c++17 -fsyntax-only -x c++ - 2> /dev/null; then
  g++ $FLAGS $ROOT/tests/cpp/facade_typed_surface.cpp -o $OUT/facade_typed_surface $LIBS && $OUT/facade_typed_surface
else
  echo "urdfdom_headers not found: facade_typed_surface.cpp skipped"
fi
F=$ROOT/tests/fixtures
$OUT/eigen_caller $F/ur10_like.urdf base_link tool0
$OUT/eigen_caller $F/panda_like.urdf link0 hand
$OUT/eigen_caller $F/mixed_joints.urdf world tip
$OUT/rdyn_speed_test_eigen $F/ur10_like.urdf base_link tool0 1000 | tail -3
echo "real-Eigen build: all checks passed"
