from .base_container import BaseContainer
from .wcsph_container import WCSPHContainer
from .dfsph_container import DFSPHContainer
from .pcisph_container import PCISPHContainer
