#!/bin/bash
# The host builder (fermat_amd/csrc/fpt_bvh.cpp: thread pool, parallel partition, optimiser set-up / write-back, collapse, refit) under ThreadSanitizer and under
# AddressSanitizer + UBSan: two host threads build and refit the same scene at the same time, each with its own pool of FPT_BUILD_THREADS workers.  CPU only.
#   bash tools/sanitize_builder.sh        (round 5: no report from either on a 105 604-triangle and a 1 112-triangle scene)
set -e
cd "$(dirname "$0")/.."
O=tools/_build/sanitize; mkdir -p $O
python - << 'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from fermat_amd import scene
for name, s in (("standin", scene.bathroom_standin(0.3)), ("glossy", scene.cornell_box("CornellBox-Glossy"))):
    np.ascontiguousarray(s.vertex_indices, np.int32).tofile('tools/_build/sanitize/%s_idx.bin' % name)
    np.ascontiguousarray(s.vertex_data, np.float32).tofile('tools/_build/sanitize/%s_vtx.bin' % name)
PY
for san in thread address,undefined; do
  g++ -O1 -g -std=c++17 -Wno-psabi -fsanitize=$san -fno-sanitize-recover=undefined -ffp-contract=off -Ifermat_amd/csrc -o $O/bvh_${san%%,*} tools/sanitize_builder_main.cpp fermat_amd/csrc/fpt_bvh.cpp -lpthread
  for scn in standin glossy; do echo "== -fsanitize=$san, $scn"; FPT_BUILD_THREADS=8 $O/bvh_${san%%,*} $scn; done
done
