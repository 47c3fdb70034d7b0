"""What does ONE rank's share of a strong-scaled scene cost?  A block shaped like one z-slab of BASELINE configs[3] on 8 GPUs (C4: 100 x 250 x 160
particles -> 100 x 250 x 20 = 500,000 per rank, 12-14 cell layers) and like one slab of the 1.23 M scene on 8 ranks (81 x 190 x 10 = 153,900,
6-8 layers), run on ONE GPU without sharding: per-kernel HIP-event times and ms/step = the compute floor of a rank, to which the exchange kernels
(~20 us) and the neighbours' skew add.  Prints one JSON object per shape.
    python tools/slab_size_probe.py [--steps 200]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    sys.stdout = sys.stderr
    # (default layout: a rank's slab is thin along z, the fastest axis of the cell order, like these blocks.  SPH_AXIS_ORDER=zxy gives the
    #  stand-in of the alternative layout, SPH_SLAB_LAYOUT=slow: profiles/r04_slab_layouts.txt)
    from sph_project_amd import product as P
    shapes = {"C4 / 8 ranks (500,000 particles, 14 layers)": dict(domain_end=(6.0, 6.0, 0.56), start=(0.1, 0.1, 0.08), end=(2.1, 5.1, 0.48)),
              "C2 / 8 ranks (153,900 particles, 8 layers)": dict(domain_end=(8.5, 8.0, 0.32), start=(0.09, 0.2, 0.06), end=(1.7, 4.0, 0.26)),
              "C2 / 2 ranks (615,600 particles, 24 layers)": dict(domain_end=(8.5, 8.0, 0.96), start=(0.09, 0.2, 0.08), end=(1.7, 4.0, 0.88))}
    for name, kw in shapes.items():
        cfg = P.dam_break_scene(translation=(0.0, 0.0, 0.0), velocity=(0.0, -0.5, 0.0), **kw)
        container, solver = P.build_product(cfg, fast_math=1)
        solver.prepare()
        eng = container.engine
        names = [eng.lib.sph_kernel_name(k).decode() for k in range(19)]
        eng.step_async(20); eng.synchronize()
        eng.profile_enable(-1, True); eng.profile_reset()
        eng.step_async(10); eng.synchronize()
        table = {names[k]: eng.profile_read(k) for k in range(19)}
        eng.profile_enable(-1, False)
        eng.synchronize(); t0 = time.perf_counter()
        eng.step_async(args.steps); eng.synchronize()
        el = time.perf_counter() - t0
        n = int(container.particle_num[None])
        print(json.dumps({"shape": name, "particles": n, "grid": [int(g) for g in container.grid_num], "ms_per_step": 1e3 * el / args.steps,
                          "kernels_us": {k: round(1e3 * v[1] / v[0], 1) for k, v in table.items() if v[0]}}), file=sys.__stdout__)
        eng.close()


if __name__ == "__main__":
    main()
