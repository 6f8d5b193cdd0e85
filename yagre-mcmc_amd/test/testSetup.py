"""yagremcmc.test.testSetup -> the explicit targets and the Lotka-Volterra plug-ins
the reference's examples and tests import from there."""
from ..plugins import (GaussianTargetDensity1d, GaussianTargetDensity2d,  # noqa: F401
                       LotkaVolterraParameter, LotkaVolterraSolver, generate_synthetic_data)
