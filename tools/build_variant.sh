#!/bin/bash
# tools/build_variant.sh <name> <file.hip> [-D... flags]: rebuilds ONE device translation unit with extra flags and links it with
# the stock objects into rosdyn_amd/variants/librdyn_<name>.so (git-ignored, travels to the GPU box) for tools/kbench A/B runs.
set -e
NAME=$1; SRC=$2; shift 2
cd /root/repo/rosdyn_amd/csrc
mkdir -p ../variants _obj/var
STEM=$(basename $(basename $SRC .hip) .cpp)
OBJ=_obj/var/${NAME}_${STEM}.o
if [[ $SRC == *.cpp ]]; then
  g++ -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$@" -c $SRC -o $OBJ
else
  EXTRA=""
  if [[ $SRC == rdyn_image_part.hip || $SRC == rdyn_kernels.hip ]]; then EXTRA="-mllvm -pragma-unroll-threshold=1000000"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-cuda-compat -ffp-contract=on $EXTRA "$@" -c $SRC -o $OBJ
fi
# STOCK_OBJ=<object the variant replaces> when it is not _obj/<stem>.o (the slices of rdyn_image_part.hip: _obj/rdyn_image_na6.o ...;
# of rdyn_kernels.hip: STOCK_OBJ=_obj/rdyn_kernels_part0.o with -DRDYN_KERNELS_PART=0, ...)
OTHERS=$(ls _obj/*.o | grep -v "${STOCK_OBJ:-_obj/${STEM}.o}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/librdyn_${NAME}.so $OBJ $OTHERS
echo built ../variants/librdyn_${NAME}.so
