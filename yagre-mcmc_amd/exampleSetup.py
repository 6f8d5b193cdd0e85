"""exampleSetup (repository root of the reference) -> the linear forward-model solver."""
from .plugins import ExampleLinearModelSolver  # noqa: F401
