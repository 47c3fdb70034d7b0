"""How should the 9 candidate runs of a tile be grouped into staging groups?  (round 6)

k_nbr_pass stages three runs at a time (one x offset: "group") and walks their accepted neighbours in one merged loop per group; the four waves
of a workgroup meet at a barrier behind every group (the tile is restaged).  Two costs of a grouping, both in merged-loop trips:
  per-wave   mean over waves of  sum_g max_lanes(n_g)          -- what a wave executes (the lane permutation's job: tools/analysis/imbalance.py)
  per-WG     mean over tiles of  sum_g max_all-256-lanes(n_g)  -- what the workgroup's wall clock sees: a wave that is through with a group waits at
                                                                  the barrier for the slowest wave of its workgroup
The x-offset grouping gives a lattice particle (9, 17, 0) or (0, 17, 9) neighbours per group depending on the half of the cell it sits in: waves
sorted by x are uniform inside (26 trips) but the workgroup pays 9 + 17 + 9 = 35.  Other partitions of the 9 runs spread the centre column's 11
and the edge columns' 6 differently.  This script evaluates partitions on real particle distributions (CPU oracle states)."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial import cKDTree
from tests import helpers as H

K = lambda ox, oy: (ox + 1) * 3 + (oy + 1)
GROUPINGS = {
    "x offset (today)": [[K(-1, -1), K(-1, 0), K(-1, 1)], [K(0, -1), K(0, 0), K(0, 1)], [K(1, -1), K(1, 0), K(1, 1)]],
    "latin diag": [[K(-1, -1), K(0, 0), K(1, 1)], [K(-1, 0), K(0, 1), K(1, -1)], [K(-1, 1), K(0, -1), K(1, 0)]],
    "centre + anti-diagonal corners | x edges + corner | y edges + corner": [[K(0, 0), K(-1, 1), K(1, -1)], [K(-1, 0), K(1, 0), K(-1, -1)], [K(0, -1), K(0, 1), K(1, 1)]],
    "centre alone | x edges + diag corners | y edges + anti-diag corners (1 + 4 + 4)": [[K(0, 0)], [K(-1, 0), K(1, 0), K(-1, -1), K(1, 1)], [K(0, -1), K(0, 1), K(-1, 1), K(1, -1)]],
    "centre | 4 edges | 4 corners": [[K(0, 0)], [K(-1, 0), K(1, 0), K(0, -1), K(0, 1)], [K(-1, -1), K(1, 1), K(-1, 1), K(1, -1)]],
    "one loop over all nine (bound)": [list(range(9))],
}


def counts(pos, h):
    n = len(pos)
    cell = np.floor(pos / h).astype(np.int64)
    nx, ny, nz = cell.max(0) + 2
    lin = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
    order = np.argsort(lin, kind="stable")
    pos, cell = pos[order], cell[order]
    pairs = cKDTree(pos.astype(np.float64)).query_pairs(h * (1 - 1e-6), output_type="ndarray")
    i = np.concatenate([pairs[:, 0], pairs[:, 1]]); j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    d = cell[j] - cell[i]
    cnt = np.zeros((n, 9), np.int64)
    np.add.at(cnt, (i, (d[:, 0] + 1) * 3 + (d[:, 1] + 1)), 1)
    frac = pos / h - cell
    return cnt, frac


def analyse(pos, h, label):
    cnt, frac = counts(pos, h)
    nb = len(cnt) // 256
    cnt, frac = cnt[: nb * 256].reshape(nb, 256, 9), frac[: nb * 256].reshape(nb, 256, 3)
    keys = {"x in cell (today)": frac[:, :, 0],
            "quadrant (x half, y half), then x": np.floor(frac[:, :, 0] * 2) * 4 + np.floor(frac[:, :, 1] * 2) * 2 + frac[:, :, 0] * 0.5}
    print("%s: %d tiles, %.1f neighbours per particle" % (label, nb, cnt.sum(2).mean()))
    for gname, groups in GROUPINGS.items():
        cg = np.stack([cnt[:, :, g].sum(2) for g in groups], axis=2)          # (tiles, 256, groups)
        row = []
        for kname, key in keys.items():
            perm = np.argsort(key, axis=1, kind="stable")
            w = np.take_along_axis(cg, perm[:, :, None], axis=1).reshape(nb, 4, 64, len(groups))
            per_wave = w.max(2).sum(2).mean()
            per_wg = w.max(2).max(1).sum(1).mean()
            row.append("%s: per-wave %.1f, per-WG %.1f" % (kname, per_wave, per_wg))
        print("   %-82s %s" % (gname, " | ".join(row)))


def main():
    h = 0.04
    cfg = H.dam_break_scene(end=(0.6, 0.8, 0.6))
    _, geo, batches = H.scene_particles(cfg)
    analyse(batches[0]["pos"].astype(np.float32), h, "rest lattice")
    sim = H.build_oracle(cfg); sim.prepare()
    done = 0
    for steps in (300, 1500):
        sim.step(steps - done); done = steps
        analyse(np.array(sim.field("particle_positions"))[: sim.particle_num], h, "after %d steps" % steps)


if __name__ == "__main__":
    main()
