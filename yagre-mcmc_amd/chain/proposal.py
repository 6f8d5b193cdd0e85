from ..interfaces import ProposalMethod  # noqa: F401
