#!/bin/bash
# Attribution build of the library: conv_wino2.hip with -DW2_TIME=1 (s_memtime stamps, tools/w2_segments.py), every other object
# from the product build.  Output: build_exp/libfcdgan_w2time.so (git-ignored; travels to the GPU box with the snapshot).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/fcd_gan_pytorch_amd/csrc && make -j8 > /dev/null
mkdir -p $ROOT/build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -DW2_TIME=1 \
  -c conv_wino2.hip -o $ROOT/build_exp/conv_wino2_time.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v conv_wino2.o | grep -v "build/lab_") $ROOT/build_exp/conv_wino2_time.o \
  -o $ROOT/build_exp/libfcdgan_w2time.so
echo built $ROOT/build_exp/libfcdgan_w2time.so
