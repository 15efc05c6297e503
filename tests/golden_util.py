"""Helpers to replay tests/golden/*.npz (made by the REAL reference; oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    blob = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {"draws": {}, "out": {}, "grads": {}, "terms": {}, "loss": [None, None]}
    for k in blob.files:
        v = blob[k]
        if k.startswith("draw:"):
            g["draws"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("out:"):
            g["out"][k[4:]] = torch.from_numpy(v)
        elif k.startswith("grad:"):
            g["grads"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("term:"):
            g["terms"][k[5:]] = float(v)
        elif k.startswith("loss:"):
            g["loss"][int(k[5:])] = float(v)
        else:
            g[k] = torch.from_numpy(v) if v.ndim else v.item()
    return g


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
