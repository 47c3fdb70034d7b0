"""Lane-imbalance estimate for the neighbour pass: for waves of 64 consecutive sorted particles, compares
sum_runs max_lanes(accepted), sum_groups max_lanes(sum of 3 runs), max_lanes(total) and the mean."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.spatial import cKDTree
from tests import helpers as H

def analyse(pos, h, label):
    n = len(pos)
    cell = np.floor(pos / h).astype(np.int64)
    nx, ny, nz = cell.max(0) + 2
    lin = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
    order = np.argsort(lin, kind="stable")
    pos = pos[order]; cell = cell[order]
    tree = cKDTree(pos.astype(np.float64))
    pairs = tree.query_pairs(h * (1 - 1e-6), output_type="ndarray")
    i = np.concatenate([pairs[:, 0], pairs[:, 1]]); j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    d = cell[j] - cell[i]
    run = (d[:, 0] + 1) * 3 + (d[:, 1] + 1)
    cnt = np.zeros((n, 9), np.int64)
    np.add.at(cnt, (i, run), 1)
    nw = n // 64
    c = cnt[: nw * 64].reshape(nw, 64, 9)
    per_run = c.max(1).sum(1)
    per_grp = c.reshape(nw, 64, 3, 3).sum(3).max(1).sum(1)
    tot = c.sum(2).max(1)
    mean = c.sum(2).mean(1)
    # static in-block permutation: the 256 particles of a workgroup sorted once by their x position inside the cell
    nb = n // 256
    fx = (pos[:, 0] / h - cell[:, 0])[: nb * 256].reshape(nb, 256)
    cg = cnt[: nb * 256].reshape(nb, 256, 3, 3).sum(3)
    perm = np.argsort(fx, axis=1, kind="stable")
    cgs = np.take_along_axis(cg, perm[:, :, None], axis=1).reshape(nb, 4, 64, 3)
    static_x = cgs.max(2).sum(2).mean()
    key2 = cg[:, :, 0] - cg[:, :, 2]
    perm2 = np.argsort(-key2, axis=1, kind="stable")
    cgs2 = np.take_along_axis(cg, perm2[:, :, None], axis=1).reshape(nb, 4, 64, 3)
    static_c = cgs2.max(2).sum(2).mean()
    # static permutation by TOTAL accepted count (known at the end of the mask-building pass)
    perm3 = np.argsort(-cg.sum(2), axis=1, kind="stable")
    cgs3 = np.take_along_axis(cg, perm3[:, :, None], axis=1).reshape(nb, 4, 64, 3)
    static_t = cgs3.max(2).sum(2).mean()
    # ... and by (frac-x bucket of 4, total count): x position first, count inside
    key4 = np.floor(fx * 4).astype(np.int64) * 1000 - cg.sum(2)
    perm4 = np.argsort(key4, axis=1, kind="stable")
    cgs4 = np.take_along_axis(cg, perm4[:, :, None], axis=1).reshape(nb, 4, 64, 3)
    static_xt = cgs4.max(2).sum(2).mean()
    print(f"   static sort by total count {static_t:.1f}; by (x quarter, count) {static_xt:.1f}; ideal (mean per wave) {cg.reshape(nb, 4, 64, 3).sum(3).mean():.1f}")
    # per-group sort by count (what a per-group hand-over achieves)
    srt = -np.sort(-cg, axis=1)
    pg_sorted = srt.reshape(nb, 4, 64, 3).max(2).sum(2).mean()
    print(f"   static sort by frac-x {static_x:.1f}; by c(-1)-c(+1) {static_c:.1f}; per-group sorted {pg_sorted:.1f}")
    # 128-wide alternatives: pairs of lanes (2 particles per lane) -> balance by pairing lane l with l+64
    print(f"{label}: n={n} mean nbrs {mean.mean():.1f} | trips/wave: per-run {per_run.mean():.1f}  per-group {per_grp.mean():.1f}  all-9 {tot.mean():.1f}")

r = 0.01; h = 4 * r
cfg = H.dam_break_scene(end=(0.6, 0.8, 0.6))
_, geo, batches = H.scene_particles(cfg)
analyse(batches[0]["pos"].astype(np.float32), h, "rest lattice")
for steps in (300, 1500):
    sim = H.build_oracle(cfg); sim.prepare(); sim.step(steps)
    analyse(np.array(sim.field("particle_positions"))[: sim.particle_num], h, f"after {steps} steps")
