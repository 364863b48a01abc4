"""Learning-rate schedules of the reference trainer family (yolox_24p/utils/lr_scheduler.py:9-205), host side.

``LRScheduler(name, lr, iters_per_epoch, total_epochs, **kwargs).update_lr(iters)`` with the reference's five names
(``cos``, ``warmcos``, ``yoloxwarmcos``, ``yoloxsemiwarmcos``, ``multistep``), keyword arguments and defaults.  The
value is a python float computed in double precision with the reference's operation order, so the two agree to the
last bit (tests/test_schedule.py pins them against values produced by the reference itself).  The captured training
step takes the value through a device-resident hyper-parameter block (ep24.train.TrainStep.set_lr): changing the
rate never re-captures a graph.
"""
import math


def _half_cosine(num, den):
    return 0.5 * (1.0 + math.cos(math.pi * num / den))


class LRScheduler:
    def __init__(self, name, lr, iters_per_epoch, total_epochs, **kwargs):
        self.lr = lr
        self.iters_per_epoch = iters_per_epoch
        self.total_epochs = total_epochs
        self.total_iters = iters_per_epoch * total_epochs
        self.__dict__.update(kwargs)                      # lr_scheduler.py:29: options become attributes
        self.lr_func = self._get_lr_func(name)

    def update_lr(self, iters):
        return self.lr_func(iters)

    # ------------------------------------------------------------------ one closure per schedule name
    def _get_lr_func(self, name):
        build = {"cos": self._cos, "warmcos": self._warmcos, "yoloxwarmcos": self._yolox_warmcos,
                 "yoloxsemiwarmcos": self._yolox_semi_warmcos, "multistep": self._multistep}.get(name)
        if build is None:
            raise ValueError("Scheduler version {} not supported.".format(name))
        return build()

    def _cos(self):                                       # lr_scheduler.py:95-98
        base, total = self.lr, self.total_iters
        return lambda it: base * _half_cosine(it, total)

    def _warmcos(self):                                   # lr_scheduler.py:101-117, linear warm-up from 1e-6
        base, total = self.lr, self.total_iters
        warm = self.iters_per_epoch * self.warmup_epochs
        start = getattr(self, "warmup_lr_start", 1e-6)

        def f(it):
            if it <= warm:
                return (base - start) * it / float(warm) + start
            return base * _half_cosine(it - warm, total - warm)
        return f

    def _yolox_warmcos(self):                             # lr_scheduler.py:120-148, quadratic warm-up, floor at the end
        base, total = self.lr, self.total_iters
        warm = self.iters_per_epoch * self.warmup_epochs
        tail = self.iters_per_epoch * self.no_aug_epochs
        start = getattr(self, "warmup_lr_start", 0)
        floor = base * getattr(self, "min_lr_ratio", 0.2)

        def f(it):
            if it <= warm:
                return (base - start) * pow(it / float(warm), 2) + start
            if it >= total - tail:
                return floor
            return floor + 0.5 * (base - floor) * (1.0 + math.cos(math.pi * (it - warm) / (total - warm - tail)))
        return f

    def _yolox_semi_warmcos(self):                        # lr_scheduler.py:151-197
        base, total = self.lr, self.total_iters
        start = getattr(self, "warmup_lr_start", 0)
        floor = base * getattr(self, "min_lr_ratio", 0.2)
        warm = self.iters_per_epoch * self.warmup_epochs
        tail = self.iters_per_epoch * self.no_aug_epochs
        normal = self.iters_per_epoch * self.semi_epoch
        semi = self.iters_per_epoch_semi * (self.total_epochs - self.semi_epoch - self.no_aug_epochs)
        ipe, ipe_semi = self.iters_per_epoch, self.iters_per_epoch_semi

        def f(it):
            if it <= warm:
                return (base - start) * pow(it / float(warm), 2) + start
            if it >= normal + semi:
                return floor
            if it <= normal:
                phase = it - warm
            else:                                         # semi-supervised epochs advance the cosine at their own rate
                phase = normal - warm + (it - normal) * ipe * 1.0 / ipe_semi
            return floor + 0.5 * (base - floor) * (1.0 + math.cos(math.pi * phase / (total - warm - tail)))
        return f

    def _multistep(self):                                 # lr_scheduler.py:84-91, :200-204
        base = self.lr
        marks = [int(self.total_iters * m / self.total_epochs) for m in self.milestones]
        gamma = getattr(self, "gamma", 0.1)

        def f(it):
            v = base
            for m in marks:
                v *= gamma if it >= m else 1.0
            return v
        return f
