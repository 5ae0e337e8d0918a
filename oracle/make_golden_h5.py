#!/usr/bin/env python3
"""Generate tests/golden/g10_feature_file.h5 + g10_feature_items.npz -- a real HDF5 file in the reference's feature layout and what the
reference's own item-read statements return for it (VL-T5/src/vqa_data_memory.py:166-187).

TEST INFRASTRUCTURE.  Runs ONLY in the build container (needs /root/reference and libhdf5).  The file is written by the HDF5 C library
(`vqacl_amd.hdf5_io.write_feature_file`: the library h5py wraps) from seeded arrays; the statements of `VQAFineTuneDataset.__getitem__`
from `feats = np.zeros(...)` to `boxes.clamp_(...)` are taken out of the reference's source with `ast` and executed as they are with
`f` bound to `vqacl_amd.hdf5_io.H5File(path)` -- the h5py calls they make (`f[name].read_direct`, `f[name][()]`, KeyError) are the
interface that class restates.  Nothing of the reference's text is written out: the fixture holds the file and the tensors those
statements produced; the test regenerates the seeded arrays itself, so the file's content is pinned twice.

The file carries no object timestamps (`write_feature_file(track_times=False)`): running this script again reproduces it byte for byte.

Usage:  python oracle/make_golden_h5.py
"""
import ast
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, ROOT)

IMAGES = (("COCO_val2014_000000000042", 640, 480), ("458752", 500, 375), ("9", 333, 500))


def seeded_items(seed=10, n_boxes=36, feat_dim=2048):
    """The arrays the fixture file holds (tests/test_hdf5_cpu.py regenerates them from the same seed)."""
    rng = np.random.default_rng(seed)
    items = {}
    for img_id, w, h in IMAGES:
        feats = np.maximum(rng.standard_normal((n_boxes, feat_dim)).astype(np.float32), 0) * np.float32(1.5)
        boxes = np.sort(rng.random((n_boxes, 4)).astype(np.float32), axis=1)
        boxes[:, (0, 2)] *= w
        boxes[:, (1, 3)] *= h
        boxes[0] = (0.0, 0.0, w, h)
        items[img_id] = dict(features=feats, boxes=boxes, img_w=np.int64(w), img_h=np.int64(h),
                             obj_id=rng.integers(0, 1600, n_boxes), obj_conf=rng.random(n_boxes).astype(np.float32))
    return items


def main():
    from vqacl_amd import hdf5_io as H
    path = os.path.join(OUT, "g10_feature_file.h5")
    items = seeded_items()
    H.write_feature_file(path, items)
    src_path = os.path.join(REF, "VL-T5/src/vqa_data_memory.py")
    tree = ast.parse(open(src_path).read())
    cls = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == "VQAFineTuneDataset")
    getitem = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__getitem__")
    body = None
    for n in ast.walk(getitem):
        if isinstance(n, ast.If) and "use_vision" in ast.unparse(n.test):
            body = n.body
    a = next(i for i, s in enumerate(body) if isinstance(s, ast.Assign) and ast.unparse(s).startswith("feats = np.zeros"))
    b = next(i for i, s in enumerate(body) if "clamp_" in ast.unparse(s))
    mod = ast.Module(body=body[a:b + 1], type_ignores=[])
    code = compile(ast.fix_missing_locations(mod), src_path, "exec")
    f = H.H5File(path)
    out = {}
    for img_id, w, h in IMAGES:
        env = {"f": f, "img_id": img_id, "np": np, "torch": torch, "self": types.SimpleNamespace(n_boxes=36), "datum": None,
               "out_dict": {}}
        exec(code, env)
        assert torch.equal(env["out_dict"]["vis_feats"], torch.from_numpy(items[img_id]["features"]))
        out[f"{img_id}/vis_feats_sum"] = env["out_dict"]["vis_feats"].double().sum().numpy()
        out[f"{img_id}/vis_feats_head"] = env["out_dict"]["vis_feats"][:2, :8].numpy()
        out[f"{img_id}/boxes"] = env["boxes"].numpy()
    f.close()
    np.savez(os.path.join(OUT, "g10_feature_items.npz"), **out)
    # round 6: the same first image re-packed the way `create_dataset(..., compression="gzip", shuffle=True, chunks=(9, 512))` stores it
    # (chunked + shuffle + deflate), with an IEEE-half and a float64 copy of a feature slice beside it -- storage forms the reference's own
    # extraction scripts do not produce (they write `grp[name] = array`: contiguous, uncompressed) but re-packed feature files do; the
    # read path must return the same item for them.  (No h5py in this image: written through libhdf5's own filter pipeline; h5dump is
    # the independent reader in tests/test_hdf5_cpu.py.)
    first = IMAGES[0][0]
    packed = dict(items[first])
    packed["features_f16"] = items[first]["features"][:, :64].astype(np.float16)
    packed["features_f64"] = items[first]["features"][:4, :16].astype(np.float64)
    gz = os.path.join(OUT, "g11_feature_file_gzip.h5")
    H.write_feature_file(gz, {first: packed}, compression="gzip", compression_opts=4, shuffle=True, chunks=(9, 512))
    with H.H5File(gz) as g:
        env = {"f": g, "img_id": first, "np": np, "torch": torch, "self": types.SimpleNamespace(n_boxes=36), "datum": None, "out_dict": {}}
        exec(code, env)
        assert torch.equal(env["out_dict"]["vis_feats"], torch.from_numpy(items[first]["features"]))
        assert np.array_equal(env["boxes"].numpy(), out[f"{first}/boxes"]), "the reference's item read returns the same item from the re-packed file"
    print("g11_feature_file_gzip.h5:", os.path.getsize(gz), "bytes")
    print("g10_feature_file.h5:", os.path.getsize(path), "bytes;", len(IMAGES), "images; HDF5", ".".join(map(str, H.library_version())))


if __name__ == "__main__":
    main()
