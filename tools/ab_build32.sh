#!/bin/bash
# A/B variants of the 32-query-tile kernels: csrc/flow32.hip is rebuilt per variant, everything else (csrc/bsdfd.hip included)
# is compiled once.   tools/ab_build32.sh NAME "EXTRA HIPCC FLAGS" [NAME2 "FLAGS2" ...]  ->  build_ab/lib_NAME.so
# Run with BSDFD_TILE=32 tools/ab_run.sh ROUNDS "--only disk8,disk4,sph8" NAME ...
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build_ab"; mkdir -p "$OUT"
CS="$ROOT/bsdf_diffusion_sampling_amd/csrc"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I $ROOT/include"
for tu in bsdfd wavefront encoding measured bucket clock; do
  if [ ! -f "$OUT/$tu.o" ] || [ "$CS/$tu.hip" -nt "$OUT/$tu.o" ] || [ "$CS/flow_dev.h" -nt "$OUT/$tu.o" ] || [ "$ROOT/include/bsdfd.h" -nt "$OUT/$tu.o" ]; then
    hipcc $COMMON -c "$CS/$tu.hip" -o "$OUT/$tu.o" &
  fi
done
wait
while [ $# -gt 0 ]; do
  name="$1"; flags="$2"; shift 2
  ( d="$OUT/tmp_$name"; rm -rf "$d"; mkdir -p "$d"; cd "$d"
    hipcc $COMMON $flags -save-temps=obj -c "${FLOW32_SRC:-$CS/flow32.hip}" -o "$d/flow32.o" 2> "$d/err.txt" || { echo "FAILED $name"; grep -m5 error "$d/err.txt"; exit 1; }
    cp "$d"/*-hip-amdgcn-amd-amdhsa-gfx950.s "$OUT/flow32_$name.s"; cp "$d/flow32.o" "$OUT/flow32_$name.o"
    hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/bsdfd.o" "$OUT/flow32_$name.o" "$OUT/wavefront.o" "$OUT/encoding.o" "$OUT/measured.o" "$OUT/bucket.o" "$OUT/clock.o" -o "$OUT/lib_$name.so" && \
    echo "built $name ($flags): $(grep -A14 'name:.*flow_kernel32' $OUT/flow32_$name.s | grep -E 'vgpr_count|private_segment_fixed' | tr -s ' ' | tr '\n' ' ')"
    rm -rf "$d" ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
