#!/bin/bash
# Build A/B variants of libbsdfd.so into build_ab/ (travels to the GPU box, git-ignored):
#   tools/ab_build.sh NAME "EXTRA HIPCC FLAGS" [NAME2 "FLAGS2" ...]
# The side translation units (csrc/flow32.hip included) are compiled once; only csrc/bsdfd.hip is rebuilt per variant.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build_ab"; mkdir -p "$OUT"
CS="$ROOT/bsdf_diffusion_sampling_amd/csrc"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I $ROOT/include"
for tu in flow32 wavefront encoding measured bucket clock; do
  if [ ! -f "$OUT/$tu.o" ] || [ "$CS/$tu.hip" -nt "$OUT/$tu.o" ] || [ "$CS/flow_dev.h" -nt "$OUT/$tu.o" ] || [ "$ROOT/include/bsdfd.h" -nt "$OUT/$tu.o" ]; then
    hipcc $COMMON -c "$CS/$tu.hip" -o "$OUT/$tu.o" &
  fi
done
wait
while [ $# -gt 0 ]; do
  name="$1"; flags="$2"; shift 2
  ( hipcc $COMMON $flags -c "$CS/bsdfd.hip" -o "$OUT/bsdfd_$name.o" && \
    hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/bsdfd_$name.o" "$OUT/flow32.o" "$OUT/wavefront.o" "$OUT/encoding.o" "$OUT/measured.o" "$OUT/bucket.o" "$OUT/clock.o" -o "$OUT/lib_$name.so" && echo "built $name ($flags)" ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
