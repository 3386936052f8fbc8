"""npz <-> torch helpers for the committed golden vectors (tests/golden/*.npz).

numpy has no bfloat16: bf16 tensors are stored as their uint16 bit pattern under the key
"<name>__bf16".  Python ints / floats / lists are stored as 0-d / 1-d arrays.
"""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def save_case(name: str, **items) -> str:
    out = {}
    for key, val in items.items():
        if isinstance(val, torch.Tensor):
            val = val.detach().cpu().contiguous()
            if val.dtype == torch.bfloat16:
                out[key + "__bf16"] = val.view(torch.int16).numpy().view(np.uint16)
            else:
                out[key] = val.numpy()
        else:
            out[key] = np.asarray(val)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    return path


def load_case(name: str) -> dict:
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    data = np.load(path, allow_pickle=False)
    out = {}
    for key in data.files:
        arr = data[key]
        if key.endswith("__bf16"):
            out[key[: -len("__bf16")]] = torch.from_numpy(arr.view(np.int16).copy()).view(torch.bfloat16)
        elif arr.ndim == 0:
            out[key] = arr.item()
        else:
            out[key] = torch.from_numpy(arr.copy())
    return out


def list_cases(prefix: str):
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith(prefix) and f.endswith(".npz"))
