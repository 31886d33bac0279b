#!/bin/bash
# build a variant of libdcarl_hip.so with extra -D flags:  tools/build_variant.sh tools/ab/libX.so -DFOO ...
# (the source list is dcarl_amd/build.py's; use it with DCARL_HIP_LIB=<out> — an explicitly chosen build skips the build-id check)
out=$1; shift
d=$(mktemp -d)
for f in $(python -c "import sys; sys.path.insert(0, '.'); from dcarl_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))"); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans "$@" -c dcarl_amd/csrc/$f.hip -o $d/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -ldl -o $out && rm -rf $d && echo built $out
