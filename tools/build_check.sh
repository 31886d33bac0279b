#!/bin/bash
# build both variants of the library in-tree and FAIL LOUDLY unless both are current (run before every gpurun call: a stale or
# unbuildable tree makes the GPU box try to compile, and burns the call)
cd "$(dirname "$0")/.."
python dcarl_amd/build.py --all > /tmp/dcarl_build.log 2>&1
rc=$?
python - <<'PY'
import sys
sys.path.insert(0, '.')
from dcarl_amd import build
bad = [v or 'product' for v in ('', 'ab') if build.needs_build(v)]
if bad:
    print("BUILD FAILED / STALE:", bad)
    print(open('/tmp/dcarl_build.log').read()[-3000:])
    sys.exit(1)
print("build ok:", build.build_info('').get('build_id'), build.build_info('ab').get('build_id'))
PY
