from ..plugins import (CovarianceMatrix, DenseCovarianceMatrix, DiagonalCovarianceMatrix,  # noqa: F401
                       IIDCovarianceMatrix)
