"""``DataPrefetcher`` under the reference's package name (yolox_24p/data/data_prefetcher.py:8-51)."""
import _path  # noqa: F401
from ep24.input import DataPrefetcher  # noqa: F401
