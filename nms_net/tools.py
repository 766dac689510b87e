"""nms_net.tools (reference nms_net/tools.py:11-35)."""
from gossipnet_amd.tools import Timer  # noqa: F401
