from ..plugins import EvaluationStatus  # noqa: F401
