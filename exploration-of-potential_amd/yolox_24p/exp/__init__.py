from .base_exp import BaseExp
from .build import get_exp
from .yolox_base import Exp
