from ..plugins import BayesianRegressionModel  # noqa: F401
