#!/bin/bash
# Same-box A/B libraries of the ONLINE kernels only: every other translation unit is compiled once and shared.
#   tools/build_trace_variants.sh A: Q:-DDCARL_QC=1 T:-DDCARL_ATOMIC_A=1 QT:-DDCARL_QC=1,-DDCARL_ATOMIC_A=1
# -> tools/ab/lib<NAME>.so (use with DCARL_HIP_LIB=...; an explicitly chosen build skips the build-id check)
set -e
cd "$(dirname "$0")/.."
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans"
TRACE_TUS=${TRACE_TUS:-trace_nwave_f32}
d=${DCARL_VARIANT_OBJ:-/tmp/dcarl_variant_obj}; mkdir -p $d tools/ab
ALL=$(python -c "import sys; sys.path.insert(0, '.'); from dcarl_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))")
for f in $ALL; do
  case " $TRACE_TUS " in *" $f "*) continue;; esac
  if [ ! -f $d/$f.o ] || [ dcarl_amd/csrc/$f.hip -nt $d/$f.o ] || [ include/dcarl.h -nt $d/$f.o ] || [ dcarl_amd/csrc/common.h -nt $d/$f.o ]; then
    $HIPCC -c dcarl_amd/csrc/$f.hip -o $d/$f.o 2> >(grep -v "argument unused" >&2) &
  fi
done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  flags=${flags//,/ }
  mkdir -p $d/$name
  for f in $TRACE_TUS; do
    sf=$(python -c "import sys; sys.path.insert(0, '.'); from dcarl_amd.build import SOURCE_FLAGS; print(' '.join(SOURCE_FLAGS.get('$f.hip', [])))")   # the unit's own shipped flags
    case " $flags " in *"-amdgpu-sched-strategy"*) sf="";; esac                        # (a variant that sets the strategy itself replaces them)
    $HIPCC $sf $flags -c dcarl_amd/csrc/$f.hip -o $d/$name/$f.o 2> >(grep -v "argument unused" >&2) &
  done
done
wait
OTHER=""
for f in $ALL; do case " $TRACE_TUS " in *" $f "*) ;; *) OTHER="$OTHER $d/$f.o";; esac; done
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHER $d/$name/*.o -ldl -o tools/ab/lib$name.so && echo built tools/ab/lib$name.so
done
