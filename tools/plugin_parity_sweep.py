#!/usr/bin/env python3
"""Plugin-level accuracy of ONE build of libbsdfd.so ($BSDFD_LIB_PATH) on all 77 shipped plugin weight sets at N queries each,
both tilings, against the pinned fp64 oracle, p99 with bootstrap 95 % intervals (tests/parity77.py):

    python tools/plugin_parity_sweep.py [--n 65536] [--out gpurun_out/plugin_parity_77sets.json] [--only chm_orange]

The committed record: profiles/r06_plugin_parity_77sets.json."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plugin_parity_77sets.json"))
    ap.add_argument("--only", default=None, help="substring filter on the set names")
    ap.add_argument("--tiles", default="32,16")
    ap.add_argument("--precision", default="split3", help="split3 (the product) | f32: the exact-fp32-MFMA validation kernels (16-query tiles "
                                                          "only: pass --tiles 16; the tile label then names the row, not the kernel)")
    a = ap.parse_args()
    import parity77 as P
    from bsdf_diffusion_sampling_amd import _lib
    sets = [s for s in P.all_sets() if a.only is None or a.only in s[0]]
    rec = P.run(n=a.n, sets=sets, tiles=tuple(int(t) for t in a.tiles.split(",")), precision=a.precision)
    rec["summary"]["precision"] = a.precision
    rec["summary"]["library"] = _lib.lib().bsdfd_version().decode()
    rec["summary"]["kernel_source_sha256"] = _lib.kernel_source_sha256()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec["summary"]))


if __name__ == "__main__":
    main()
