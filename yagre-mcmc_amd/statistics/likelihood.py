from ..plugins import AdditiveGaussianNoiseLikelihood  # noqa: F401
