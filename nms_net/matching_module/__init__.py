"""nms_net.matching_module (reference nms_net/matching_module/__init__.py:9-13): `detection_matching`."""
from gossipnet_amd.matching_module import detection_matching  # noqa: F401

__all__ = 'detection_matching'
