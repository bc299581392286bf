#!/bin/bash
# builds the working tree's library as a named variant next to the in-tree one (for tools/ab_bench.sh): bash tools/build_variant.sh <name> [hipcc flags]
N=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include extendedrtirtmodeling.jl_amd/csrc/ertirt.hip "$@" -o extendedrtirtmodeling.jl_amd/libertirt_$N.so && echo extendedrtirtmodeling.jl_amd/libertirt_$N.so
