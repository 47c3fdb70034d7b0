from .DFSPH import DFSPHSolver
from .WCSPH import WCSPHSolver
from .PCISPH import PCISPHSolver
