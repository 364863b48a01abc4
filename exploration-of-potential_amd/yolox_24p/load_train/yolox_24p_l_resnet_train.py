"""BASELINE config 4: YOLOX-l-24p neck and head on the reference's resnet backbone (models/darknet.py; the switch is the stock
tree's YOLOPAFPN(backbone_type), yolox/models/yolo_pafpn.py:31-38)."""
from exp import Exp as MyExp


class Exp(MyExp):
    def __init__(self):
        super(Exp, self).__init__()
        self.depth = 1.00
        self.width = 1.00
        self.backbone_type = "resnet"
        self.num_classes = 80
        self.max_epoch = 300
        self.L1_epoch = 100
        self.exp_name = "yolox_24p_l_resnet"
