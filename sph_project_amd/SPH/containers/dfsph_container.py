from sph_project_amd import _lib as F
from .base_container import BaseContainer, _FieldView


class DFSPHContainer(BaseContainer):
    """dfsph_container.py:13-17 of the reference: alpha, kappa, kappa_v, rho*, D(rho)/Dt."""
    METHOD = "dfsph"

    def __init__(self, config, GGUI=False, **engine_opts):
        super().__init__(config, GGUI, **engine_opts)
        self.particle_dfsph_alphas = _FieldView(self, F.F_DFSPH_ALPHA)
        self.particle_dfsph_kappa = _FieldView(self, F.F_DFSPH_KAPPA)
        self.particle_dfsph_kappa_v = _FieldView(self, F.F_DFSPH_KAPPA_V)
        self.particle_densities_star = _FieldView(self, F.F_DENSITY_STAR)
        self.particle_densities_derivatives = _FieldView(self, F.F_DENSITY_DERIV)
