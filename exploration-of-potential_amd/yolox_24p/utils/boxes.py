"""``utils.boxes`` of the reference (yolox_24p/utils/boxes.py), the functions of the 24p path: ``circle_inter`` (:102-163, not in the
reference's ``__all__``, so it is reached as ``utils.boxes.circle_inter``), ``bboxes_iou`` (:166-243) and ``postprocess`` (:29-99)."""
import _path  # noqa: F401
from ep24.infer import postprocess  # noqa: F401
from ep24.loss import bboxes_iou, circle_inter  # noqa: F401

__all__ = ["postprocess", "bboxes_iou"]
