"""BASELINE target: YOLOX-l-24p (depth = width = 1.0, the Exp defaults, exp/yolox_base.py:16-17)."""
from exp import Exp as MyExp


class Exp(MyExp):
    def __init__(self):
        super(Exp, self).__init__()
        self.depth = 1.00
        self.width = 1.00
        self.num_classes = 80
        self.max_epoch = 300
        self.L1_epoch = 100
        self.exp_name = "yolox_24p_l"
