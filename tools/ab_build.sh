#!/bin/bash
# usage: tools/ab_build.sh <name> <extra hipcc flags...>   -- a second build of the library with other compile-time options
# (e.g. tools/ab_build.sh approx_alpha -DED3_EXACT_ALPHA=0) into e-d3dgs_amd/csrc/variants/libed3dgs_hip_<name>.so; run anything
# against it with ED3DGS_LIB_PATH=<that file>.  The variants directory is git-ignored (*.so) and travels to the GPU box.
set -e
name=$1; shift
cd /root/repo/e-d3dgs_amd/csrc
mkdir -p variants/obj_$name
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-function $*"
for f in api preprocess binning render_forward render_backward preprocess_backward deform deform_deep activations filter3d knn integrate stats; do
  extra=""; [ $f = preprocess ] && extra="-ffp-contract=off"
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c $f.hip -o variants/obj_$name/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libed3dgs_hip_$name.so variants/obj_$name/*.o
rm -rf variants/obj_$name
ls -la variants/libed3dgs_hip_$name.so
