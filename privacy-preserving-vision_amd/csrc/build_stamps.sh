#!/bin/bash
# diagnostic build of libppv_hip.so with -DPPV_STAMPS into ../lib_stamps (tools/conv_timeline.py); -fgpu-rdc: g_stamps is shared
set -e
cd "$(dirname "$0")"
mkdir -p ../lib_stamps
make -j8 >/dev/null
for f in *.hip; do
  o=../lib_stamps/${f%.hip}.o
  case $f in conv_gemm.hip|conv_stream.hip|conv_wgrad_stem.hip) ;; *) cp ../lib/${f%.hip}.o $o ;; esac
done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-result -DPPV_STAMPS -fgpu-rdc -I../../include -c conv_gemm.hip -o ../lib_stamps/conv_gemm.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-result -DPPV_STAMPS -fgpu-rdc -I../../include -c conv_stream.hip -o ../lib_stamps/conv_stream.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -Wno-unused-result -DPPV_STAMPS -fgpu-rdc -I../../include -c conv_wgrad_stem.hip -o ../lib_stamps/conv_wgrad_stem.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fgpu-rdc -o ../lib_stamps/libppv_hip.so ../lib_stamps/*.o
