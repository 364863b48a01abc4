"""Offline 24-point label generation with the reference's ``Polygon_24`` surface (datasets/2+24_labels_create.py):

    cd exploration-of-potential_amd/yolox_24p
    python datasets/labels_create_24p.py --json instances_train2017.json --images train2017 --out COCO_24p_label

``json_anno_process`` walks the COCO annotations exactly as the reference does (:122-199: crowds, areas below one pixel and
annotations whose image file is missing are skipped; class ids are re-indexed to 0..79; box centre = bbox corner + half
size), but the ray casting, the convex-hull area of the acceptance filter and nothing else run on the GPU for a whole
batch of annotations at a time (ep24.labels24 -> ep24_ray24 / ep24_hull_area24).  ``save_24r_to_txt`` writes the
reference's files (``%d`` + 50 x ``%0.4f`` in "Cord" mode, + 26 in "Radius" mode).  Mask decoding needs pycocotools,
which this image does not ship: the class raises ImportError with that message when it is constructed without it.
"""
import argparse
import json
import os
from pathlib import Path

import numpy as np

import _path  # noqa: F401
from ep24 import labels24

# the 80 "thing" category ids of COCO 2017 in ascending order -> contiguous class index (:36-51)
COCO_IDS = [i for i in range(1, 91) if i not in (12, 26, 29, 30, 45, 66, 68, 69, 71, 83)]


class Polygon_24:
    def __init__(self, mode="Cord", json_label_pth=None, image_data_pth=None, new_label_pth="./COCO_24p_label", batch=1024):
        try:
            from pycocotools.coco import COCO
        except ImportError as e:
            raise ImportError("labels_create_24p needs pycocotools to decode COCO masks (annToMask)") from e
        self.mode = mode
        self.json_label_pth, self.image_data_pth, self.new_label_pth = json_label_pth, image_data_pth, new_label_pth
        self.batch = batch
        self.coco = COCO(self.json_label_pth)
        self.json_dict = self.load_label_json()
        self.label_dict_cord24, self.label_dict_radius = {}, {}
        self.coco_id2idx = {str(cid): i for i, cid in enumerate(COCO_IDS)}

    def load_label_json(self):
        with open(self.json_label_pth, "r") as f:
            return json.load(f)

    def rotation_for_24p(self, center_x, center_y, mask):
        return labels24.rotation_for_24p(center_x, center_y, mask)

    def _flush(self, pending, area_t_low, area_t_high):
        if not pending:
            return
        names, cls, centres, masks, areas = zip(*pending)
        pts, rad = labels24.rays_batch(list(masks), centres)
        hull = labels24.hull_areas(pts)
        keep, cord, radius = labels24.label_rows(cls, centres, [m.shape for m in masks], pts, rad, hull, areas, area_t_low,
                                                 area_t_high)
        for j, i in enumerate(np.nonzero(keep)[0]):
            self.label_dict_cord24[names[i]].append(cord[j])
            self.label_dict_radius[names[i]].append(radius[j])
        del pending[:]

    def json_anno_process(self, area_t_low=0.5, area_t_high=1.5):
        sizes = {im["id"]: (im["height"], im["width"]) for im in self.json_dict.get("images", [])}
        pending = []
        for anno in self.json_dict["annotations"]:
            name = str(anno["image_id"]).zfill(12)
            self.label_dict_cord24.setdefault(name, [])
            self.label_dict_radius.setdefault(name, [])
            if anno["iscrowd"] or anno["area"] < 1:
                continue
            if not os.path.exists(Path(self.image_data_pth) / Path(name + ".jpg")):
                continue
            mask = self.coco.annToMask(anno)                  # uint8 [H,W]; the json's height / width = the image's
            assert tuple(mask.shape) == tuple(sizes.get(anno["image_id"], mask.shape))
            cx = anno["bbox"][0] + anno["bbox"][2] / 2
            cy = anno["bbox"][1] + anno["bbox"][3] / 2
            pending.append((name, self.coco_id2idx[str(anno["category_id"])], (cx, cy), mask, anno["area"]))
            if len(pending) >= self.batch:
                self._flush(pending, area_t_low, area_t_high)
        self._flush(pending, area_t_low, area_t_high)
        return self.label_dict_cord24, self.label_dict_radius

    def save_24r_to_txt(self):
        label_dict = self.label_dict_cord24 if self.mode == "Cord" else self.label_dict_radius
        os.makedirs(self.new_label_pth, exist_ok=True)
        width = 51 if self.mode == "Cord" else 27
        for name, rows in label_dict.items():
            labels24.save_rows(Path(self.new_label_pth) / Path(name + ".txt"), np.array(rows).reshape(-1, width))


if __name__ == "__main__":
    ap = argparse.ArgumentParser("24-point label generation")
    ap.add_argument("--json", required=True)
    ap.add_argument("--images", required=True)
    ap.add_argument("--out", default="./COCO_24p_label")
    ap.add_argument("--mode", default="Cord", choices=["Cord", "Radius"])
    a = ap.parse_args()
    polygon = Polygon_24(a.mode, a.json, a.images, a.out)
    polygon.json_anno_process()
    polygon.save_24r_to_txt()
