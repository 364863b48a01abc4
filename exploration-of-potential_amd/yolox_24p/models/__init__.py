"""Same names as the reference's ``models`` package (models/__init__.py:2-8); implementations are the MI355X path."""
import _path  # noqa: F401
from ep24.loss import IOUloss, Loss_Function  # noqa: F401
from ep24.nn import (BaseConv, Bottleneck, CSPDarknet, CSPLayer, DWConv, Focus, SPPBottleneck, YOLOPAFPN, YOLOX,  # noqa: F401
                     YOLOXHead)
