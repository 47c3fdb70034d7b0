from .config_builder import SimConfig
