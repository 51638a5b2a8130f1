#!/bin/bash
# Builds variants/liblo_amd_<name>.so from the current tree with ONE source file replaced (experiments: several kernels of
# the same name timed on one GPU box; tools/mb_*.py load a variant when LO_LIB_VARIANT names it).
# usage: tools/build_variant.sh <name> [<file.hip> <replacement>]
set -e
cd "$(dirname "$0")/../linear_operator_amd/csrc"
name=$1
mkdir -p build ../../variants
objs=""  # (the replacement and its object live in build/_variant_*: remove them after an experiment)
for f in *.hip; do
  o=build/${f%.hip}.o
  if [ "${2:-}" = "$f" ]; then
    cp "$3" build/_variant_$f
    flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I. -I../../include"
    case $f in lo_pivchol.hip|lo_pivchol_onchip.hip) flags="$flags -ffp-contract=off";; esac
    case $f in lo_cg_onchip4.hip|lo_pivchol_onchip.hip|lo_cg_lockstep.hip|lo_rspace.hip|lo_rspace3.hip|lo_solve_fused*.hip) flags="$flags -fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc $flags -x hip -c build/_variant_$f -o build/_variant_${f%.hip}.o
    o=build/_variant_${f%.hip}.o
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/liblo_amd_$name.so $objs
ls -la ../../variants/liblo_amd_$name.so
