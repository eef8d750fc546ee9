#!/bin/bash
# Build a variant of the library next to the default one, for A/B runs on one GPU box (tools/ab.py):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSIPX_F64_VEC=4]   ->  setintersectionprojection.jl_amd/libsipx_<name>.so
# Select it with SIPX_LIBRARY=<path> (host.py) or pass <name> to tools/ab.py.  `git stash; tools/build_variant.sh prev; git stash pop`
# gives the committed state as the baseline of an uncommitted change.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/setintersectionprojection.jl_amd/csrc
out=$(mktemp -d)
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Wno-pass-failed -Wno-unused-result $*"
pids=()
for f in kernels_cds.hip kernels_sets.hip kernels_multi.hip kernels_proj.hip ext_proj.hip comm.cpp engine.cpp api.cpp; do
  hipcc $FLAGS -x hip -I"$src" -c "$src/$f" -o "$out/${f%.*}.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/setintersectionprojection.jl_amd/libsipx_$name.so" "$out"/*.o -L/opt/rocm/lib -lhipfft -lrocsolver -lrocblas -ldl
rm -rf "$out"
echo "built libsipx_$name.so"
