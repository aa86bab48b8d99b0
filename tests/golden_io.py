"""Loader for the golden fixtures in tests/golden (made by tests/golden/make_goldens.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=""):
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    return [os.path.splitext(os.path.basename(f))[0] for f in files]


def load(name):
    """Return a dict of numpy arrays; decodes the q8 value encoding and the sparse
    grad_value encoding of G5 into dense float64 arrays."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    d = {k: z[k] for k in z.files}
    if "value_q8" in d:
        d["value"] = d.pop("value_q8").astype(np.float64) * float(d.pop("value_scale"))
    if "grad_value_rows" in d:
        C = d["value"].shape[-1]
        gv = np.zeros((d["value"].size // C, C), dtype=np.float64)
        gv[d.pop("grad_value_rows")] = d.pop("grad_value_vals")
        d["grad_value"] = gv.reshape(d["value"].shape)
    if "mask_size" in d:
        d["mask_size"] = int(d["mask_size"])
    return d


# (operator-level fixtures; G7 / G9 are module-level, G8 layer-level)
BOX = [n for n in names("G") if "_box_" in n and not n.startswith(("G7", "G8", "G9"))]
INST = [n for n in names("G") if "_inst_" in n and not n.startswith(("G7", "G8", "G9"))]
