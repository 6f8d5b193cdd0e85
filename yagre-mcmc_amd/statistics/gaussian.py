from ..plugins import Gaussian, GaussianDensity  # noqa: F401
