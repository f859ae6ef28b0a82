"""Evaluation output layout of the sampler (second half): what ``inf4eval.py`` writes after each batch of
``fashion_generation(..., return_dict=False)`` so that the reference's ``Evaluation/`` scripts find the same tree.

    <gen_save_path>/images/<uid>/<oid>/<i>.jpg      generated item i of the outfit (PIL images from the HIP VAE decode)
    <gen_save_path>/images/<uid>/<oid>/all.jpg      GOR: the generated items on a ceil(sqrt(n))-wide white grid
    <gen_save_path>/images/<uid>/<oid>/grd.jpg      FITB: the ground-truth outfit on the same grid
    <gen_save_path>.npy                             {uid: {oid: {cates, full_cates, outfits, image_paths}}} (object array)
    <grd_save_path>.npy                             {uid: {oid: {outfits, image_paths}}}: ground-truth paths of the generated slots

Follows DiFashion/inf4eval.py:774-827 (save_batch_outputs) and :829-842 (merge_and_save_images); host-side file plumbing
only -- the images themselves come from ``DiFashion.fashion_generation`` on the HIP path.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Sequence

import numpy as np
import torch


def image_grid(images: Sequence):
    """Square-ish contact sheet: ceil(sqrt(n)) columns AND rows of the first image's cell size, white background, row-major
    (inf4eval.py:829-842)."""
    from PIL import Image
    side = math.ceil(math.sqrt(len(images)))
    w, h = images[0].width, images[0].height
    sheet = Image.new("RGB", (w * side, h * side), color=(255, 255, 255))
    for n, img in enumerate(images):
        sheet.paste(img, ((n % side) * w, (n // side) * h))
    return sheet


def merge_and_save_images(images: Sequence, save_path: str) -> None:
    image_grid(images).save(save_path)


def _outfit_dir(gen_save_path: str, uid, oid) -> str:
    d = os.path.join(gen_save_path, "images", str(uid), str(oid))
    os.makedirs(d, exist_ok=True)
    return d


def save_batch_outputs(all_outputs: Dict, all_grds: Dict, outputs: Dict, gen_save_path: str, task: str, all_img_folder_path: str,
                       all_image_paths, test_grd_dict: Dict, save_grd: bool = True):
    """Write one batch of generated outfits and fold its records into the running dictionaries (same arguments, return value
    and in-place effects as inf4eval.py:774: the ``images`` entry of every record is replaced by ``image_paths``; a
    (uid, oid) already present in ``all_outputs`` / ``all_grds`` keeps its first record)."""
    from PIL import Image
    for uid, per_user in outputs.items():
        for oid, rec in per_user.items():
            folder = _outfit_dir(gen_save_path, uid, oid)
            images = rec.pop("images")
            if task == "GOR":
                merge_and_save_images(images, os.path.join(folder, "all.jpg"))
            paths = []
            for i, img in enumerate(images):
                paths.append(os.path.join(folder, f"{i}.jpg"))
                img.save(paths[-1])
            rec["image_paths"] = paths
            all_outputs.setdefault(uid, {}).setdefault(oid, rec)
            if task == "FITB":
                truth = [Image.open(os.path.join(all_img_folder_path, all_image_paths[iid])) for iid in test_grd_dict[oid]["outfits"]]
                merge_and_save_images(truth, os.path.join(folder, "grd.jpg"))
    if save_grd:
        for uid, per_user in outputs.items():
            for oid, rec in per_user.items():
                if oid in all_grds.setdefault(uid, {}):
                    continue
                truth = test_grd_dict[oid]["outfits"]
                # the ground-truth item of each generated slot: the position of the slot's category in the full outfit
                paths = [os.path.join(all_img_folder_path, all_image_paths[truth[torch.where(rec["full_cates"] == cate)[0]]])
                         for cate in rec["cates"]]
                all_grds[uid][oid] = {"outfits": truth, "image_paths": paths}
    return all_outputs, all_grds


def write_eval_records(gen_save_path: str, all_outputs: Dict, grd_save_path: str | None = None, all_grds: Dict | None = None) -> None:
    """The per-batch checkpoint of the running dictionaries (inf4eval.py:752-754): 0-d object arrays, ``.npy`` appended by numpy."""
    np.save(gen_save_path, np.array(all_outputs))
    if grd_save_path is not None and all_grds is not None:
        np.save(grd_save_path, np.array(all_grds))
