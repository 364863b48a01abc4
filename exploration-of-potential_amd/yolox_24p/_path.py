"""Makes ``ep24`` importable when the entry points are run from this directory (cwd = yolox_24p/, as the
reference's README does: ``cd yolox_24p && python train_24p.py -f load_train/yolox_24p_train.py -b 20 -l 0.01``)."""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
