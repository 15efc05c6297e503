"""Helpers to replay tests/golden/*.npz (made by the REAL reference; oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    blob = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {"draws": {}, "out": {}, "grads": {}, "terms": {}, "loss": [None, None], "post": {}}
    for k in blob.files:
        v = blob[k]
        if k.startswith("draw:"):
            g["draws"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("out:"):
            g["out"][k[4:]] = torch.from_numpy(v)
        elif k.startswith("grad:"):
            g["grads"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("post:"):
            g["post"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("term:"):
            g["terms"][k[5:]] = float(v)
        elif k.startswith("loss:"):
            g["loss"][int(k[5:])] = float(v)
        else:
            g[k] = v if v.dtype.kind in "US" else (torch.from_numpy(v) if v.ndim else v.item())
    return g


def parity_log(msg, fname="flash_parity.txt"):
    """append a measured-parity line to gpurun_out/<fname> (copied to profiles/ at the end of a round) and print it"""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, fname), "a") as f:
        f.write(msg + "\n")
    print(msg, flush=True)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_sample_case(name="sample_lcm4"):
    """the few-step sampler fixture (oracle/make_golden.py::make_sample_golden)"""
    blob = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {"lora": {}, "noises": []}
    for k in blob.files:
        v = blob[k]
        if k.startswith("lora:"):
            g["lora"][k[5:]] = torch.from_numpy(v)
        elif k.startswith("lcm_noise:"):
            g["noises"].append((int(k.split(":")[1]), torch.from_numpy(v)))
        else:
            g[k] = torch.from_numpy(v) if v.ndim else v.item()
    g["noises"] = [n for _, n in sorted(g["noises"], key=lambda t: t[0])]
    return g


def sampler_models_from_golden(g):
    """the seeded tiny teacher / student of the fixtures, with the fixture's LoRA values loaded into the student"""
    from oracle.golden_cases import build_models
    teacher, student, disc = build_models()
    named = dict(student.named_parameters())
    for n, v in g["lora"].items():
        named[n].data.copy_(v)
    return teacher, student, disc
