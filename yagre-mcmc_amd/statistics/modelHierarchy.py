from ..plugins import BayesianRegressionModelHierarchy  # noqa: F401
