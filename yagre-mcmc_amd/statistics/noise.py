from ..plugins import CentredGaussianNoise  # noqa: F401
