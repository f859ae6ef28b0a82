"""Evaluation output layout (second half): difashion_amd.evalio.save_batch_outputs against the record of the
REAL reference functions (tests/golden/make_golden_evalio.py -> evalio.npz) on the same synthetic batches."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.dirname(__file__))
from helpers_data import evalio_case  # noqa: E402

from difashion_amd import evalio  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "evalio.npz"), allow_pickle=False)


def _tree(root):
    out = []
    for d, _, files in os.walk(root):
        out += [os.path.relpath(os.path.join(d, f), root) for f in files]
    return sorted(out)


@pytest.mark.parametrize("task", ["FITB", "GOR"])
def test_save_batch_outputs_matches_reference_record(tmp_path, task):
    d = str(tmp_path)
    case = evalio_case(d, task)
    all_out, all_grd = {}, {}
    for batch in case["batches"]:
        ret = evalio.save_batch_outputs(all_out, all_grd, batch, case["gen"], task, case["img_root"], case["paths"], case["grd"], True)
        assert ret[0] is all_out and ret[1] is all_grd
        assert all("images" not in r and "image_paths" in r for u in batch.values() for r in u.values())     # in-place effect
    assert _tree(case["gen"]) == list(GOLD[f"{task}_tree"])
    for key in GOLD.files:
        if key.startswith(f"{task}_px_"):
            got = np.asarray(Image.open(os.path.join(case["gen"], key[len(task) + 4:]))).astype(int)
            want = GOLD[key].astype(int)
            assert got.shape == want.shape and np.abs(got - want).max() <= 2, key      # same PIL encoder: equal up to its version
    flat_o = [f"{uid}|{oid}|{sorted(r)}|{[int(c) for c in r['cates']]}|{[int(c) for c in r['full_cates']]}|"
              f"{[int(c) for c in r['outfits']]}|{[os.path.relpath(p, d) for p in r['image_paths']]}"
              for uid in all_out for oid, r in all_out[uid].items()]
    flat_g = [f"{uid}|{oid}|{sorted(r)}|{[int(c) for c in r['outfits']]}|{[os.path.relpath(p, d) for p in r['image_paths']]}"
              for uid in all_grd for oid, r in all_grd[uid].items()]
    assert flat_o == list(GOLD[f"{task}_outputs"])
    assert flat_g == list(GOLD[f"{task}_grds"])


def test_image_grid_geometry():
    imgs = [Image.new("RGB", (10, 12), color=(i, i, i)) for i in range(5)]
    sheet = evalio.image_grid(imgs)
    assert sheet.size == (30, 36)                                   # ceil(sqrt(5)) = 3 columns and rows
    px = np.asarray(sheet)
    assert (px[12:24, 10:20] == 4).all() and (px[24:, :] == 255).all() and (px[12:24, 20:] == 255).all()


def test_write_eval_records_round_trip(tmp_path):
    rec = {7: {100: {"image_paths": ["a.jpg"], "cates": [1]}}}
    evalio.write_eval_records(str(tmp_path / "gen"), rec, str(tmp_path / "grd"), {7: {100: {"outfits": [1, 2]}}})
    back = np.load(str(tmp_path / "gen.npy"), allow_pickle=True).item()      # how Evaluation/ reads it
    assert back == rec and os.path.exists(str(tmp_path / "grd.npy"))
