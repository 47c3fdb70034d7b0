#!/usr/bin/env python
"""GGUI-free driver with the command line, loop arithmetic and on-disk outputs of the reference's
run_simulation.py (:13-44 argument + interval arithmetic, :116-155 loop, :137-144 PLY export).

    python sph_project_amd/run_simulation.py --scene_file data/scenes/high_fluid_wcsph.json [--max_steps N]

Frames go to {scene_name}_output/{cnt:06}/particle_object_{id}.ply (ASCII PLY, x y z per vertex -- the
layout Taichi's PLYWriter.export_ascii produces and surface_reconstruction.py / splashsurf consume).
Rigid bodies: mesh_object_{id}.obj per frame with exportObj (:146-150).  PNG frames (exportFrame) need the reference's
GGUI window and are not produced."""
import argparse
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)  # makes `import SPH` resolve to sph_project_amd/SPH (drop-in)

import numpy as np  # noqa: E402
from SPH.utils import SimConfig  # noqa: E402
from SPH.containers import DFSPHContainer, WCSPHContainer, PCISPHContainer  # noqa: E402
from SPH.fluid_solvers import DFSPHSolver, WCSPHSolver, PCISPHSolver  # noqa: E402


PLY_COMMENT = "created by PLYWriter"   # taichi.tools.PLYWriter's default `comment`


def write_ply_ascii(path, pos):
    """What `PLYWriter(num_vertices=n); add_vertex_pos(x, y, z); export_ascii(path)` leaves on disk (run_simulation.py:139-144).
    Layout restated from Taichi's python/taichi/tools/ply.py (>= 1.6; the package is absent here, so the layout is NOT compared with
    a Taichi run): `print_header` writes "ply", "format ascii 1.0", "comment created by PLYWriter", "element vertex N", one
    "property float x|y|z" line per channel (`add_vertex_pos` registers the three channels as "float" and casts the data to
    np.float32) and "end_header"; `export_ascii` then appends one line per vertex, every value as `str(np.float32)` -- the shortest
    digits that round-trip -- FOLLOWED by a blank, i.e. "x y z \n".  `ndarray.astype(str)` produces exactly those strings."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    if not os.environ.get("SPH_PLY_PYTHON"):
        # the same bytes from C++ behind the C-ABI (csrc/sph_export.hpp): 0.2 s instead of ~5 s for the 1.23 M particles of a frame
        from sph_project_amd import _lib
        rc = _lib.load().sph_write_ply_ascii(os.fsencode(path), pos.ctypes.data, pos.shape[0])
        if rc != 0:
            raise OSError(f"sph_write_ply_ascii({path!r}) failed ({rc})")
        return
    with open(path, "w") as f:
        f.write(f"ply\nformat ascii 1.0\ncomment {PLY_COMMENT}\n")
        f.write(f"element vertex {pos.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        if pos.shape[0]:
            f.write(" \n".join(map(" ".join, pos.astype(str).tolist())) + " \n")


def read_ply_ascii(path):
    """Vertex positions of an ASCII PLY as (n, 3) float32 (what surface_reconstruction.py / splashsurf read back)."""
    with open(path) as f:
        assert f.readline().strip() == "ply"
        n, props = None, []
        for line in f:
            tok = line.split()
            if tok[:2] == ["element", "vertex"]:
                n = int(tok[2])
            elif tok[:1] == ["property"]:
                props.append(tok[-1])
            elif tok[:1] == ["end_header"]:
                break
        data = np.loadtxt(f, dtype=np.float32, ndmin=2) if n else np.zeros((0, len(props)), np.float32)
    assert data.shape == (n, len(props)), (data.shape, n, props)
    return data[:, [props.index(k) for k in ("x", "y", "z")]]


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--scene_file", default="", help="scene file")
    parser.add_argument("--max_steps", type=int, default=None, help="stop early (not in the reference)")
    parser.add_argument("--output_dir", default=None, help="default: {scene_name}_output in the cwd (reference behaviour)")
    args = parser.parse_args(argv)
    scene_path = args.scene_file
    config = SimConfig(scene_file_path=scene_path)
    scene_name = scene_path.split("/")[-1].split(".")[0]

    fps = config.get_cfg("fps")
    if fps is None:
        fps = 60
    frame_time = 1.0 / fps
    output_interval = int(frame_time / config.get_cfg("timeStepSize"))
    total_time = config.get_cfg("totalTime")
    if total_time is None:
        total_time = 10.0
    total_rounds = int(total_time / config.get_cfg("timeStepSize"))
    if config.get_cfg("outputInterval"):
        output_interval = config.get_cfg("outputInterval")
    output_ply = config.get_cfg("exportPly")
    output_obj = config.get_cfg("exportObj")
    out_dir = args.output_dir or f"{scene_name}_output"
    os.makedirs(out_dir, exist_ok=True)

    method = config.get_cfg("simulationMethod")
    table = {"dfsph": (DFSPHContainer, DFSPHSolver), "wcsph": (WCSPHContainer, WCSPHSolver),
             "pcisph": (PCISPHContainer, PCISPHSolver)}
    if method not in table:
        raise NotImplementedError(f"Simulation method {method} not implemented")
    container = table[method][0](config, GGUI=False)
    solver = table[method][1](container)
    print(f"Simulation method: {method}")
    solver.prepare()

    cnt = 0
    limit = total_rounds if args.max_steps is None else min(total_rounds, args.max_steps)
    limit = max(limit, 1)   # the reference's loop steps once before it looks at the round count
    t0 = time.perf_counter()
    t_export, frames = 0.0, 0
    while cnt < limit:
        # run_simulation.py:126-153 steps once, writes a frame if the count of steps BEFORE this one is a multiple of the
        # interval, then counts.  Same frames here, but the steps between two frames go to the device in one call.
        wants_frame = output_ply or output_obj
        nxt = cnt if cnt % output_interval == 0 else cnt + output_interval - cnt % output_interval   # next count that gets a frame
        if not wants_frame or nxt >= limit:
            solver.advance(limit - cnt)
            cnt = limit
            break
        solver.advance(nxt - cnt + 1)
        cnt = nxt
        container.engine.synchronize()
        te = time.perf_counter()
        wrote = False
        if output_ply:
            os.makedirs(f"{out_dir}/{cnt:06}", exist_ok=True)
            for f_body_id in container.object_id_fluid_body:
                write_ply_ascii(f"{out_dir}/{cnt:06}/particle_object_{f_body_id}.ply", container.dump(obj_id=f_body_id)["position"])
                wrote = True
        if output_obj:   # run_simulation.py:146-150
            os.makedirs(f"{out_dir}/{cnt:06}", exist_ok=True)
            for r_body_id in container.object_id_rigid_body:
                if "mesh" not in container.object_collection[r_body_id]:   # body given as pre-voxelised points only
                    continue
                with open(f"{out_dir}/{cnt:06}/mesh_object_{r_body_id}.obj", "w") as f:
                    f.write(container.object_collection[r_body_id]["mesh"].export(file_type="obj"))
                wrote = True
        frames += 1 if wrote else 0
        t_export += time.perf_counter() - te
        cnt += 1
    container.engine.synchronize()
    dt = time.perf_counter() - t0
    # frame export (ASCII PLY of every fluid particle, OBJ of every rigid mesh: run_simulation.py:137-150) is host file I/O and can
    # dwarf the simulation -- 1.3 s per frame for the 1.23 M particles of final_scene0.json -- so it is reported apart from the steps
    print(f"Simulation Finished: {cnt} steps, {container.particle_num[None]} particles, {1e3 * (dt - t_export) / cnt:.3f} ms/step "
          f"(+ {t_export:.2f} s writing {frames} frame(s): {1e3 * dt / cnt:.3f} ms/step all in)")
    return container, solver


if __name__ == "__main__":
    main()
