"""nms_net.dataset (reference nms_net/dataset.py:17-112; the TF queue Prefetcher :115-140 has no counterpart)."""
from gossipnet_amd.dataset import ShuffledDataset, TestDataset, load_roi  # noqa: F401
