from ..plugins import SolverInterface  # noqa: F401
