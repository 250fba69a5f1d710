"""Stand-in for NVIDIA apex (absent here): amp at opt-level O0 is the identity."""
from . import amp  # noqa: F401
