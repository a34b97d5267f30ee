#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...]: builds tools/_variants/libcrd_NAME.so (same sources, extra -D flags) for A/B
# timing against the in-tree library (CRD_LIBRARY=tools/_variants/libcrd_NAME.so); the objects go to a private directory.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/tools/_variants; mkdir -p $OUT/obj_$NAME
cd $ROOT/crdmodel_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I/opt/rocm/include $*"
for f in crd_host.cpp crd_io.cpp crd_context.cpp crd_halo.cpp crd_steppers.cpp crd_trace.cpp crd_kernel_table.cpp; do /opt/rocm/bin/hipcc $FLAGS -DCRD_NO_KERNEL_TABLE -x hip -c $f -o $OUT/obj_$NAME/${f%.*}.o & done
for f in crd_kernels.hip crd_fused.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $OUT/obj_$NAME/${f%.*}.o & done
/opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -c crd_fused_f32.hip -o $OUT/obj_$NAME/crd_fused_f32.o &  # (as the Makefile builds it)
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libcrd_$NAME.so $OUT/obj_$NAME/*.o -L/opt/rocm/lib -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
rm -rf $OUT/obj_$NAME
echo $OUT/libcrd_$NAME.so
