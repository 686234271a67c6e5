"""SMPL model ingestion without chumpy.

The reference loads `SMPL_{NEUTRAL,FEMALE,MALE}.pkl` through chumpy (`serialization.py:1-39`,
`smpl_layer.py:37-63`) and keeps seven arrays.  The official pickles contain chumpy `Ch` objects and a
scipy sparse `J_regressor`; here the chumpy classes are replaced at unpickling time by a shim that
only keeps the wrapped ndarray, so neither chumpy nor OpenCV is needed.  `.npz` files with the same
keys are accepted too (what `save_npz` writes).
"""
import pickle

import numpy as np


class _ChShim:
    """Stands in for chumpy.ch.Ch (and subclasses) while unpickling: keeps the state dict; `.r` is the array."""

    def __init__(self, *a, **k):
        self._state = {}

    def __setstate__(self, state):
        self._state = state if isinstance(state, dict) else {"x": state}

    @property
    def r(self):
        for key in ("x", "_x", "a"):
            if key in self._state:
                return np.asarray(self._state[key])
        raise ValueError("chumpy object without a stored array")

    def __array__(self, dtype=None, copy=None):
        a = self.r
        return a.astype(dtype) if dtype is not None else a


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return _ChShim
        return super().find_class(module, name)


def _arr(v):
    if isinstance(v, _ChShim):
        return v.r
    if hasattr(v, "toarray"):      # scipy sparse J_regressor (smpl_layer.py:52)
        return np.asarray(v.toarray())
    return np.asarray(v)


def load_smpl_model(path):
    """-> dict(v_template[V,3], shapedirs[V,3,10], posedirs[V,3,207], J_regressor[24,V], weights[V,24],
    parents[24], model_betas[10], f[F,3]) as float32 / int arrays."""
    if str(path).endswith(".npz"):
        dd = dict(np.load(path, allow_pickle=False))
    else:
        with open(path, "rb") as fh:
            dd = _Unpickler(fh, encoding="latin1").load()
    out = {}
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
        out[k] = np.ascontiguousarray(_arr(dd[k]), dtype=np.float32)
    # the official pickles carry 300 shape directions in some releases; the layer uses what is there
    if "parents" in dd:
        parents = np.asarray(dd["parents"]).reshape(-1)
    else:
        parents = np.asarray(_arr(dd["kintree_table"]))[0]          # smpl_layer.py:60-62
    parents = parents.astype(np.int64).copy()
    parents[0] = -1                                                   # stored as 2**32-1 in the pickles
    out["parents"] = parents.astype(np.int32)
    nb = out["shapedirs"].shape[-1]
    out["model_betas"] = (np.asarray(_arr(dd["betas"]), np.float32).reshape(-1) if "betas" in dd
                          else np.zeros(nb, np.float32))              # serialization.py:21-22
    if "f" in dd:
        out["f"] = np.asarray(_arr(dd["f"])).astype(np.int64)
    return out


def save_npz(path, model):
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in model.items()})
