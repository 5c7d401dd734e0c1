#!/usr/bin/env python3
"""Host-only: what the GATHER maps of a 4-node mesh look like under the library's numbering (no device).

  python tools/gather_stats.py [nx ny nz] [--quadratic]     (10-node elements of, else) corner tetrahedra of the reference's TetGen deck, nx x ny x nz copies

Prints chunks, rows per chunk, element evaluations per element and the histogram of chunk lengths: the figures the
numbering of an unstructured mesh is judged by (csrc/renumber.cpp) before a device sees it."""
import gzip
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fea-large_amd"))
import feahip  # noqa: E402
import mesh  # noqa: E402


def main():
    quadratic = "--quadratic" in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = tuple(int(a) for a in argv[:3]) if len(argv) >= 3 else (2, 2, 2)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "brick_fine.sexp")
        with gzip.open(os.path.join(ROOT, "tests", "golden", "decks", "brick_fine.sexp.gz"), "rb") as src, open(p, "wb") as dst:
            shutil.copyfileobj(src, dst)
        deck = feahip.Deck.load(p)
    lin = deck if quadratic else mesh.corner_tets(deck)
    tiled = mesh.tiled(lin, reps)
    t0 = time.time()
    perm, renumbered = feahip.host_numbering(tiled.elements, tiled.nodes)
    t1 = time.time()
    st, hist = feahip.host_gather_stats(perm[tiled.elements], len(tiled.nodes))
    st["numbering_s"] = round(t1 - t0, 2)
    st["maps_s"] = round(time.time() - t1, 2)
    st["renumbered"] = renumbered
    print(st)
    nz = np.nonzero(hist)[0]
    print("chunks by rows:", {int(k): int(hist[k]) for k in nz})


if __name__ == "__main__":
    main()
