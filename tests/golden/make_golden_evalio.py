#!/usr/bin/env python3
"""Golden record of the evaluation output layout (second half) from the REAL reference functions.  Build
container only (needs /root/reference); the committed ``evalio.npz`` is what travels.

``DiFashion/inf4eval.py`` cannot be imported here (diffusers / accelerate drivers at module level), so the two functions are
taken out of its syntax tree at generation time and executed unmodified in a namespace holding the names they use
(os, math, torch, PIL.Image).  Nothing of the reference's text is stored: the fixture holds the synthetic inputs' seeds, the
file tree the functions produced, the decoded contact sheets and the returned dictionaries.
"""
import ast
import math
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers_data import evalio_case  # noqa: E402


def reference_functions():
    src = open("/root/reference/DiFashion/inf4eval.py").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("save_batch_outputs", "merge_and_save_images")]
    assert len(keep) == 2
    ns = {"os": os, "math": math, "torch": torch, "Image": Image}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "inf4eval.py", "exec"), ns)
    return ns["save_batch_outputs"]


def tree_of(root):
    out = []
    for d, _, files in os.walk(root):
        out += [os.path.relpath(os.path.join(d, f), root) for f in files]
    return sorted(out)


def main():
    save = reference_functions()
    rec = {}
    for task in ("FITB", "GOR"):
        with tempfile.TemporaryDirectory() as d:
            case = evalio_case(d, task)
            all_out, all_grd = {}, {}
            for batch in case["batches"]:
                all_out, all_grd = save(all_out, all_grd, batch, case["gen"], task, case["img_root"], case["paths"], case["grd"], True)
            rec[f"{task}_tree"] = np.array(tree_of(case["gen"]))
            for rel in rec[f"{task}_tree"]:
                if os.path.basename(rel) in ("all.jpg", "grd.jpg", "0.jpg"):
                    rec[f"{task}_px_{rel}"] = np.asarray(Image.open(os.path.join(case["gen"], rel)))
            flat_o, flat_g = [], []
            for uid in all_out:
                for oid in all_out[uid]:
                    r = all_out[uid][oid]
                    flat_o.append(f"{uid}|{oid}|{sorted(r)}|{[int(c) for c in r['cates']]}|{[int(c) for c in r['full_cates']]}|"
                                  f"{[int(c) for c in r['outfits']]}|{[os.path.relpath(p, d) for p in r['image_paths']]}")
            for uid in all_grd:
                for oid in all_grd[uid]:
                    r = all_grd[uid][oid]
                    flat_g.append(f"{uid}|{oid}|{sorted(r)}|{[int(c) for c in r['outfits']]}|{[os.path.relpath(p, d) for p in r['image_paths']]}")
            rec[f"{task}_outputs"], rec[f"{task}_grds"] = np.array(flat_o), np.array(flat_g)
    np.savez_compressed(os.path.join(HERE, "evalio.npz"), **rec)
    print("wrote evalio.npz with", len(rec), "arrays")


if __name__ == "__main__":
    main()
