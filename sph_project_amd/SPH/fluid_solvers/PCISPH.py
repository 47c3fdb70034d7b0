from ..containers import PCISPHContainer
from .base_solver import BaseSolver


class PCISPHSolver(BaseSolver):
    """PCISPH.py of the reference (predictive-corrective pressure loop)."""

    def __init__(self, container: PCISPHContainer):
        super().__init__(container)
        self.max_iterations = 1000
        self.eta = 0.001
