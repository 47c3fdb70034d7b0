"""JSON scene loader with the interface of the reference's SPH/utils/config_builder.py:5-44
(SimConfig.get_cfg and the four get_* list accessors; absent keys give None / [])."""
import json


class SimConfig:
    _LISTS = {"get_rigid_bodies": "RigidBodies", "get_rigid_blocks": "RigidBlocks",
              "get_fluid_bodies": "FluidBodies", "get_fluid_blocks": "FluidBlocks"}

    def __init__(self, scene_file_path=None, config=None, verbose=False) -> None:
        if config is not None:
            self.config = config
        else:
            with open(scene_file_path, "r") as fh:
                self.config = json.load(fh)
        if verbose:
            print(self.config)

    def get_cfg(self, name, enforce_exist=False):
        section = self.config["Configuration"]
        if name in section:
            return section[name]
        assert not enforce_exist, f"missing configuration key {name}"
        return None

    def _list(self, key):
        return self.config.get(key, [])

    def get_rigid_bodies(self):
        return self._list("RigidBodies")

    def get_rigid_blocks(self):
        return self._list("RigidBlocks")

    def get_fluid_bodies(self):
        return self._list("FluidBodies")

    def get_fluid_blocks(self):
        return self._list("FluidBlocks")
