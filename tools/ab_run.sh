#!/bin/bash
# Interleaved A/B of the variants in build_ab/ on the GPU box:  tools/ab_run.sh ROUNDS "ab.py args" NAME [NAME ...]
# -> gpurun_out/ab_<timestamp>.jsonl (one line per variant per round)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ROUNDS="$1"; ARGS="$2"; shift 2
mkdir -p "$ROOT/gpurun_out"; OUTF="$ROOT/gpurun_out/ab_$(date +%H%M%S).jsonl"
for r in $(seq 1 "$ROUNDS"); do
  for n in "$@"; do
    extra=""; [ "$r" = 1 ] && extra="--acc"
    # NAME@16 / NAME@32: the same library with the 16- / 32-query-tile kernels selected (BSDFD_TILE)
    lib="${n%@*}"; tile=""; [ "$lib" != "$n" ] && tile="${n#*@}"
    ( [ -n "$tile" ] && export BSDFD_TILE="$tile"
      BSDFD_LIB_PATH="$ROOT/build_ab/lib_$lib.so" timeout 600 python3 "$ROOT/tools/ab.py" --tag "$n" $ARGS $extra | tail -1 >> "$OUTF" )
  done
done
python3 "$ROOT/tools/ab_summary.py" "$OUTF"
