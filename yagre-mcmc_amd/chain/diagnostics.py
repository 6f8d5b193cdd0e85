"""yagremcmc.chain.diagnostics -> ensemble diagnostics."""
from ..diagnostics import AcceptanceRateDiagnostics, DummyDiagnostics, FullDiagnostics  # noqa: F401
