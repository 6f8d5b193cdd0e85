from ..interfaces import ChainDiagnostics  # noqa: F401
