from ..plugins import EvaluationStatus, ForwardModel  # noqa: F401
