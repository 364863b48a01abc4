"""Experiment base class: same surface as the reference's ``exp/base_exp.py:14-80`` (abstract factories,
tabulated ``__repr__``, ``merge`` of ``[key, value, ...]`` overrides with type coercion)."""
import ast
import pprint
from abc import ABCMeta, abstractmethod

from tabulate import tabulate


class BaseExp(metaclass=ABCMeta):
    def __init__(self):
        self.output_dir = "./YOLOX_outputs"
        self.print_interval = 100
        self.eval_interval = 10

    @abstractmethod
    def get_model(self):
        ...

    @abstractmethod
    def get_data_loader(self, batch_size):
        ...

    @abstractmethod
    def get_optimizer(self, lr):
        ...

    @abstractmethod
    def get_lr_scheduler(self, lr, iters_per_epoch, **kwargs):
        ...

    def __repr__(self):
        rows = [(str(k), pprint.pformat(v)) for k, v in vars(self).items() if not k.startswith("_")]
        return tabulate(rows, headers=["keys", "values"], tablefmt="fancy_grid")

    def merge(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            if not hasattr(self, k):
                continue                                  # only keys that already exist are updated
            cur = getattr(self, k)
            if cur is not None and type(cur) != type(v):
                try:
                    v = type(cur)(v)
                except Exception:
                    v = ast.literal_eval(v)
            setattr(self, k, v)
