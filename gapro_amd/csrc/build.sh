#!/usr/bin/env bash
# Build libgapro_hip.so in-tree for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libgapro_hip.so"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed \
  -o "${out}" "${here}/ctx.hip" "${here}/partition.hip" "${here}/svgp_fit.hip" "${here}/svgp_fit_small.hip" "${here}/svgp_fit_large.hip" "${here}/svgp_fit_cluster.hip" "${here}/labels.hip" "${here}/consumer.hip" "${here}/debug_peak.hip" "${here}/schedule.cpp"
echo "built ${out}"
if [[ "${GAPRO_BUILD_PROFILE:-0}" == "1" ]]; then
  # diagnostic build with in-kernel phase stamps; never used by the product or the tests
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-pass-failed -DGAPRO_PROFILE \
    -o "${here}/../libgapro_hip_prof.so" "${here}/ctx.hip" "${here}/partition.hip" "${here}/svgp_fit.hip" "${here}/svgp_fit_small.hip" "${here}/svgp_fit_large.hip" "${here}/svgp_fit_cluster.hip" "${here}/labels.hip" "${here}/consumer.hip" "${here}/debug_peak.hip" "${here}/schedule.cpp"
  echo "built ${here}/../libgapro_hip_prof.so"
fi
