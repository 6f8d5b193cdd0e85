from ..plugins import CovarianceMatrix, DiagonalCovarianceMatrix, IIDCovarianceMatrix  # noqa: F401
