#!/usr/bin/env bash
# Build libgapro_hip.so in-tree for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# One object per translation unit under csrc/build/ (git-ignored), compiled in parallel and only when the source or a
# header is newer than the object; GAPRO_BUILD_PROFILE=1 also builds the diagnostic library with in-kernel phase
# stamps (libgapro_hip_prof.so: never used by the product or the tests).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
srcs=(ctx.hip devmem.hip feeder.hip partition.hip svgp_fit.hip svgp_fit_small.hip svgp_fit_wave.hip svgp_fit_large.hip svgp_fit_cluster.hip labels.hip consumer.hip schedule.cpp pth_io.cc)
# libgapro_hip_debug.so: measurement / self-test entry points (include/gapro_hip_debug.h), never loaded by the product
debug_srcs=(svgp_fit_debug.hip debug_peak.hip)
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed)

build_lib() {  # $1 = object directory, $2 = output library, rest = extra flags
  local odir="$1" out="$2"; shift 2
  mkdir -p "${odir}"
  # the -D flags are part of the build's identity: a variant rebuilt with other flags must not reuse objects
  local flagstr="$*"
  if [[ ! -f "${odir}/.flags" ]] || [[ "$(cat "${odir}/.flags")" != "${flagstr}" ]]; then
    rm -f "${odir}"/*.o
    printf '%s' "${flagstr}" > "${odir}/.flags"
  fi
  local newest_hdr=0 h t
  for h in "${here}"/*.h "${here}/../../include"/*.h "${here}/build.sh"; do
    t=$(stat -c %Y "$h"); (( t > newest_hdr )) && newest_hdr=$t
  done
  local pids=() objs=() s o so
  for s in "${srcs[@]}"; do
    o="${odir}/${s%.*}.o"; objs+=("$o")
    so=$(stat -c %Y "${here}/${s}")
    # svgp_fit_small.hip / svgp_fit_debug.hip are svgp_fit.hip built again
    [[ "$s" == "svgp_fit_small.hip" || "$s" == "svgp_fit_debug.hip" ]] && { t=$(stat -c %Y "${here}/svgp_fit.hip"); (( t > so )) && so=$t; }
    if [[ ! -f "$o" ]] || (( $(stat -c %Y "$o") < so )) || (( $(stat -c %Y "$o") < newest_hdr )); then
      if [[ "$s" == *.cc ]]; then  # host-only C++ (no HIP headers): the system compiler, no device pass
        g++ -O3 -std=c++17 -fPIC -Wall -c -o "$o" "${here}/${s}" &
      else
        hipcc "${flags[@]}" "$@" -c -o "$o" "${here}/${s}" &
      fi
      pids+=($!)
    fi
  done
  local p rc=0
  for p in "${pids[@]:-}"; do [[ -n "$p" ]] && { wait "$p" || rc=1; }; done
  (( rc == 0 )) || { echo "compilation failed" >&2; exit 1; }
  hipcc --offload-arch=gfx950 -shared -fPIC -o "${out}" "${objs[@]}"
  echo "built ${out}"
}

build_lib "${here}/build/rel" "${here}/../libgapro_hip.so"
product_srcs=("${srcs[@]}")
srcs=("${debug_srcs[@]}")
build_lib "${here}/build/debug" "${here}/../libgapro_hip_debug.so"
srcs=("${product_srcs[@]}")
# experiments: GAPRO_VARIANT=name GAPRO_VARIANT_FLAGS="-DX ..." builds libgapro_hip_name.so (tools/bench_fit.py --lib)
if [[ -n "${GAPRO_VARIANT:-}" ]]; then
  # shellcheck disable=SC2086
  build_lib "${here}/build/${GAPRO_VARIANT}" "${here}/../libgapro_hip_${GAPRO_VARIANT}.so" ${GAPRO_VARIANT_FLAGS:-}
fi
if [[ "${GAPRO_BUILD_PROFILE:-0}" == "1" ]]; then
  build_lib "${here}/build/prof" "${here}/../libgapro_hip_prof.so" -DGAPRO_PROFILE
fi
