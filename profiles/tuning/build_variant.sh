#!/bin/bash
# Builds libptamd.so with extra -D tuning macros into variants/<name>.so (variants/ is not tracked; it ships to the
# GPU box with the tree).  usage, from the repository root: profiles/tuning/build_variant.sh <name> "-DPT_LEAF_LANES=8 ..."
# Macros: PT_EXTEND_REFILL_LANES, PT_CONNECT_REFILL_LANES, PT_LEAF_LANES, PT_TRAVERSE_BURST, PT_LDS_TOP_NODES, and the
# workgroup size / waves per SIMD of the traversal instances: PT_TRIANGLE_BLOCK / PT_TRIANGLE_WAVES (closest hit,
# scene without spheres), PT_TRAVERSE_BLOCK / PT_TRAVERSE_WAVES (closest hit with the sphere path), PT_CONNECT_BLOCK /
# PT_CONNECT_WAVES (any hit), PT_WIDE_BLOCK / PT_WIDE_WAVES / PT_WIDE_LDS_STACK / PT_WIDE_REFILL_LANES /
# PT_WIDE_DEFERRED / PT_WIDE_DEFER_LEAF_LANES / PT_WIDE_DEFER_KEEP_LANES / PT_WIDE_BURST (closest hit over the wide layout),
# PT_WIDE_SHADOW_REFILL_LANES / _LEAF_LANES / _KEEP_LANES / _DEFERRED / _BURST (any hit over it), PT_LIGHT_WIDE_* likewise (any hit over the wide nodes in the light's frame), PT_UNIT_BATCH (kernels.hip); PT_TOP_NODES is an environment
# variable read at upload.
mkdir -p variants/obj
for f in api bvh_build wide_bvh layout_build; do
  if [ ! -f variants/obj/$f.o ] || [ pathtracer_amd/csrc/$f.hip -nt variants/obj/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -c pathtracer_amd/csrc/$f.hip -o variants/obj/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Iinclude $2 -shared \
    -o variants/$1.so variants/obj/api.o variants/obj/bvh_build.o variants/obj/wide_bvh.o variants/obj/layout_build.o pathtracer_amd/csrc/kernels.hip
