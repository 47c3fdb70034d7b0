"""z-slab sharding: host-side planning shared by the containers, bench.py and the tests.

One process per GPU; rank r owns the global cell layers cz in [cuts[r], cuts[r+1]) (whole layers of the uniform
grid, cell size dh) plus one ghost layer per interior side.  The rules below are the same ones the
device kernels of csrc/sph_halo.hpp apply every step:
  * a particle belongs to the rank whose layer range contains cz = clamp(trunc(z / dh), 0, nz-1);
  * owned particles in the first / last layer of a slab are copied to the lower / upper neighbour as ghosts;
  * a particle that leaves its slab migrates to the neighbour; if it moved into the adjacent layer it stays
    behind as a ghost (both sides can tell from the record alone, so one message per neighbour per step suffices).
"""
from __future__ import annotations

import numpy as np


def cell_layer(z, dh, nz):
    """Global cell layer of z coordinates, in f32 like the device (`(int)(z / grid_size)`, clamped)."""
    cz = np.trunc(np.asarray(z, np.float32) / np.float32(dh)).astype(np.int64)
    return np.clip(cz, 0, nz - 1)


def plan_slabs(layer_counts, nranks, min_layers=2):
    """Cut nz layers into `nranks` contiguous slabs with balanced particle counts (every slab >= min_layers).
    Returns cuts of length nranks + 1 with cuts[0] = 0 and cuts[-1] = nz."""
    counts = np.asarray(layer_counts, dtype=np.float64)
    nz = len(counts)
    if nranks * min_layers > nz:
        raise ValueError(f"{nz} cell layers cannot host {nranks} slabs of >= {min_layers} layers")
    cum = np.concatenate([[0.0], np.cumsum(counts)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, nranks):
        target = total * r / nranks
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, nz)] - target):
            k -= 1
        k = max(k, cuts[-1] + min_layers)            # room for this slab ...
        k = min(k, nz - (nranks - r) * min_layers)   # ... and for the ones above
        cuts.append(k)
    cuts.append(nz)
    return cuts


def owner_of(cz, cuts):
    return np.searchsorted(np.asarray(cuts[1:-1]), cz, side="right")


def classify(cz, ghost, z_lo, z_hi, has_down, has_up):
    """Reference of k_halo_classify for one rank.  Returns dict of index arrays into the current particle set:
    keep_owned, keep_as_ghost (migrants that moved one layer), to_down / to_up (records, with is_ghost flags)."""
    cz = np.asarray(cz)
    live = ~np.asarray(ghost, bool)                 # last step's ghosts are dropped
    mig_dn = live & (cz < z_lo) & has_down
    mig_up = live & (cz >= z_hi) & has_up
    stay = live & ~mig_dn & ~mig_up
    copy_dn = stay & (cz == z_lo) & has_down
    copy_up = stay & (cz == z_hi - 1) & has_up & ~copy_dn
    return dict(
        keep_owned=np.nonzero(stay)[0],
        keep_as_ghost=np.nonzero((mig_dn & (cz == z_lo - 1)) | (mig_up & (cz == z_hi)))[0],
        to_down=np.nonzero(mig_dn | copy_dn)[0], to_down_ghost=copy_dn[mig_dn | copy_dn],
        to_up=np.nonzero(mig_up | copy_up)[0], to_up_ghost=copy_up[mig_up | copy_up],
    )
