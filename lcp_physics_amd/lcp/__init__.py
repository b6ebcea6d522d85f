from .lcp import LCPFunction, lcp_solve, lcp_backward  # noqa: F401
