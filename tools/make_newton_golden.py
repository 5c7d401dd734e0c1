#!/usr/bin/env python3
"""Golden vectors of the FULL load path of the reference's clamped decks (solve(), fea_solver.c:163-236).

Runs the CPU oracle (oracle/fea_oracle.c, the restated loops; direct solver as the decks ask) on
tests/golden/decks/{neohook_brick,a5_brick}.sexp with the decks' own settings for all of their
:load-increments-count 120 increments -- or to the increment at which the oracle itself hits
max-newton-count, which the reference treats as failure (fea_solver.c:225-231) -- and stores, per deck:
  its[steps]        Newton iterations of every increment
  tol[sum(its)]     <u,f> of every iteration (fea_solver.c:208-210)
  nodes[N][3]       final coordinates
  syy[E][G]         final Cauchy stress component yy at every Gauss point
This takes ~8 + ~4 minutes of one CPU core, which is why it is a committed fixture and not a test;
tests/test_gpu_parity.py compares the HIP path with it (iteration counts identical, <u,f> and displacements 1e-10).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fea-large_amd"), os.path.join(ROOT, "tests")]
import feahip                                   # noqa: E402
from oracle_binding import OracleSolver         # noqa: E402

out_dir = os.path.join(ROOT, "tests", "golden", "newton_full")
os.makedirs(out_dir, exist_ok=True)
for name in sys.argv[1:] or ("neohook_brick", "a5_brick"):
    deck = feahip.Deck.load(os.path.join(ROOT, "tests", "golden", "decks", name + ".sexp"))
    o = OracleSolver(deck)
    t0 = time.time()
    done, its, tol = o.solve(deck.load_increments_count, deck.max_newton_count, bool(deck.modified_newton),
                             deck.desired_tolerance, feahip.CHOLESKY)
    steps = min(done + 1, deck.load_increments_count)          # the failing increment ran too
    its = np.asarray(its[:steps], dtype=np.int32)
    tol = np.asarray(tol[:int(its.sum())], dtype=np.float64)
    S = o.stresses()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), done=np.int32(done), its=its, tol=tol,
                        nodes=o.nodes(), syy=S[:, :, 1, 1])
    print(f"{name}: {done} increments finished of {deck.load_increments_count}, iterations {its.tolist()}, {time.time() - t0:.0f} s")
