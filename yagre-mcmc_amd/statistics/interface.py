from ..interfaces import (BayesianModelInterface, CovarianceOperatorInterface, DensityInterface,  # noqa: F401
                          NoiseModelInterface, ParameterLawInterface)
