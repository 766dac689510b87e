"""nms_net.config (reference nms_net/config.py:10-121): the global `cfg`, `cfg_from_file`."""
from gossipnet_amd.config import AttrDict, cfg, cfg_from_file, reset_cfg  # noqa: F401
