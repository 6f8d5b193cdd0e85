from ..plugins import AdditiveGaussianNoiseLikelihood, AEMLikelihood  # noqa: F401
