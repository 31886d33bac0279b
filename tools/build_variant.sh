#!/bin/bash
# build a variant of libdcarl_hip.so with extra -D flags:  tools/build_variant.sh tools/ab/libX.so -DFOO ...
out=$1; shift
d=$(mktemp -d)
for f in abi trace trace_nwave trace_tab_f32 trace_tab_f64 bounds sampler misc rls frenet; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans "$@" -c dcarl_amd/csrc/$f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $out && rm -rf $d && echo built $out
