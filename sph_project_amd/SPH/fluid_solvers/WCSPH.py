from ..containers import WCSPHContainer
from .base_solver import BaseSolver


class WCSPHSolver(BaseSolver):
    """WCSPH.py of the reference: Tait EOS with the hard-coded stiffness 50000 / exponent 7
    (WCSPH.py:12-13; the scene keys "stiffness"/"exponent" are ignored there as well)."""

    def __init__(self, container: WCSPHContainer):
        super().__init__(container)
        self.gamma = 7.0
        self.stiffness = 50000.0
