from ..plugins import AEMNoise, CentredGaussianNoise  # noqa: F401
