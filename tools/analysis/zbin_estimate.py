#!/usr/bin/env python
"""VERDICT r05 #6, offline estimate first: would ordering the particles of a cell by z sub-bin shrink phase 1 of the density pass?

Phase 1 (sph_device.hpp phase1_mask) distance-tests, per lane and per (ox, oy) run, the candidates of the three z-cells [cz - 1, cz + 1] of
that column: m = their population, in chunks of 8 candidates with a wave-uniform trip count = max over the wave's 64 lanes of ceil(m / 8)
(at least one).  With the particles of a cell ordered by z sub-bin (B bins per cell), a lane at height fz (0..1 inside its cell) needs
  cell cz - 1: bins b >= floor(B fz)      (z_j >= z_i - h),     cell cz: all,     cell cz + 1: bins b < ceil(B fz)      (z_j <= z_i + h)
-- still one contiguous range, m' <= m.  This script takes real states of the C2 scene (from rest and in motion) off the GPU, computes m and
m' for every (particle, run) from the positions alone, and adds up the wave-uniform chunk trips of the kernel's lane = consecutive-particle
mapping for several chunk sizes.  It prices phase 1's TRIPS only (each trip = 8 tests x 8 VALU + 12 LDS reads for all 64 lanes), which is
what the experiment would save; the sort would additionally have to produce the sub-bin order (non-deterministic-order mode only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sph_project_amd import _lib as L, product as P  # noqa: E402


def trips(m, chunk):
    """wave-uniform chunk trips: m is (waves, 64, 9); bottom-tested loop -> at least one trip per run"""
    t = np.maximum(1, -(-m // chunk))
    return t.max(axis=1).sum()


def analyse(x, grid_num, dh, bins):
    n = len(x)
    c = np.minimum(np.maximum((x / np.float32(dh)).astype(np.int64), 0), np.array(grid_num) - 1)
    nx, ny, nz = (int(v) for v in grid_num)
    fz = x[:, 2].astype(np.float64) / dh - c[:, 2]
    b = np.minimum((fz * bins).astype(np.int64), bins - 1)
    lin = (c[:, 0] * ny + c[:, 1]) * nz + c[:, 2]
    # population of every (cell, bin), prefix over the z-fastest order: ranges of (cell, bin) keys are contiguous runs of the sorted order
    key = lin * bins + b
    cnt = np.bincount(key, minlength=nx * ny * nz * bins)
    pre = np.concatenate([[0], np.cumsum(cnt)])
    m = np.zeros((n, 9), np.int64)
    mz = np.zeros((n, 9), np.int64)
    k = 0
    for ox in (-1, 0, 1):
        for oy in (-1, 0, 1):
            cx, cy = c[:, 0] + ox, c[:, 1] + oy
            ok = (cx >= 0) & (cx < nx) & (cy >= 0) & (cy < ny)
            col = (np.clip(cx, 0, nx - 1) * ny + np.clip(cy, 0, ny - 1)) * nz
            z0, z1 = np.maximum(c[:, 2] - 1, 0), np.minimum(c[:, 2] + 1, nz - 1)
            lo, hi = (col + z0) * bins, (col + z1 + 1) * bins
            m[:, k] = np.where(ok, pre[hi] - pre[lo], 0)
            # sub-bin trimmed range (only where the neighbour cell exists)
            lo_b = np.where(c[:, 2] > 0, lo + np.floor(fz * bins).astype(np.int64), lo)
            hi_b = np.where(c[:, 2] < nz - 1, hi - bins + np.ceil(fz * bins).astype(np.int64), hi)
            hi_b = np.maximum(hi_b, lo_b)
            mz[:, k] = np.where(ok, pre[hi_b] - pre[lo_b], 0)
            k += 1
    pad = (-n) % 64
    if pad:
        m = np.concatenate([m, np.zeros((pad, 9), np.int64)]); mz = np.concatenate([mz, np.zeros((pad, 9), np.int64)])
    return m.reshape(-1, 64, 9), mz.reshape(-1, 64, 9)


def main():
    cfg = P.c2_scene()
    container, solver = P.build_product(cfg, fast_math=1)
    solver.prepare()
    e = container.engine
    out = {}
    for label, upto in (("from rest (step 30)", 30), ("in motion (step 2500)", 2500)):
        e.step_async(upto - int(solver.stats()["steps"])); e.synchronize()
        x = e.download(L.F_POSITION)
        row = {"particles": int(len(x))}
        for bins in (2, 4, 8):
            m, mz = analyse(x, container.grid_num, container.dh, bins)
            row["candidates_per_particle"] = float(m.sum() / len(x))
            row["bins=%d" % bins] = {"candidates_per_particle": float(mz.sum() / len(x)),
                                     **{"trips chunk %d: now %d, z-binned %d (%.1f %%)" % (ch, trips(m, ch), trips(mz, ch), 100.0 * (trips(mz, ch) / trips(m, ch) - 1.0)): None
                                        for ch in (8, 4)}}
        out[label] = row
        print(label, json.dumps(row, indent=1))
    e.close()


if __name__ == "__main__":
    main()
