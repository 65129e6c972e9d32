#!/bin/bash
# condense what tools/r06_profile.sh (+ the BSDFD_TILE=16 pass of tools/profile.sh r06_t16) left under gpurun_out/ into profiles/
set -u
cd "$(dirname "$0")/.."
python3 tools/summarize_profile.py r06 r06_disk_1Mi_T8 disk_1Mi_T8 > /dev/null
python3 tools/summarize_profile.py r06_sph r06_spherical_16Mi_T8 spherical_16Mi_T8 > /dev/null
for wl in disk_1Mi_T4 mixed_16Mi teacher_64x6_4Mi_T128 complex64_1Mi_T8; do python3 tools/summarize_profile.py r06_$wl r06_$wl $wl > /dev/null; done
python3 tools/summarize_profile.py r06_t16 r06_disk_1Mi_T8_tile16 disk_1Mi_T8@tile16 > /dev/null
python3 - <<'P'
import json
d = json.load(open("profiles/pmc_latest.json"))
print({k: (v if k == "_meta" else v["hbm_bytes_per_launch"]) for k, v in d.items()})
P
