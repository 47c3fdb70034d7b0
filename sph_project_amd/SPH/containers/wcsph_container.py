from .base_container import BaseContainer


class WCSPHContainer(BaseContainer):
    """wcsph_container.py:10 of the reference: no extra per-particle state."""
    METHOD = "wcsph"
