"""nms_net.class_weights (reference nms_net/class_weights.py:12-22)."""
from gossipnet_amd.class_weights import class_equal_weights, get_class_counts  # noqa: F401
