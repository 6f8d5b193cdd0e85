from ..plugins import ParameterVector  # noqa: F401
