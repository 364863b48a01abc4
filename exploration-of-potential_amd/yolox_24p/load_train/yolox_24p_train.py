"""The reference's shipped training Exp: YOLOX-s sized (depth 0.33, width 0.50), load_train/yolox_24p_train.py:8-19."""
from exp import Exp as MyExp


class Exp(MyExp):
    def __init__(self):
        super(Exp, self).__init__()
        self.depth = 0.33
        self.width = 0.50
        self.num_classes = 80
        self.max_epoch = 2000
        self.L1_epoch = 100
        self.data_num_workers = 4
        self.exp_name = "yolox_24p"
