"""Nested dict / list / ndarray / scalar / None  <->  flat {str: ndarray} for np.savez (no pickles in fixtures).
Dict keys may be ints or strings; both survive the round trip."""
import numpy as np


def _k(k):
    if isinstance(k, (int, np.integer)) and not isinstance(k, (bool, np.bool_)):
        return f"i{int(k)}"
    assert isinstance(k, str) and "/" not in k, k
    return f"s{k}"


def pack(tree, prefix="", out=None):
    out = {} if out is None else out
    if isinstance(tree, dict):
        out[prefix + "#d"] = np.array([_k(k) for k in tree.keys()], dtype="U")      # keeps the insertion ORDER
        for k, v in tree.items():
            pack(v, prefix + _k(k) + "/", out)
    elif isinstance(tree, (list, tuple)):
        out[prefix + "#l"] = np.array(len(tree))
        for i, v in enumerate(tree):
            pack(v, prefix + f"{i}/", out)
    elif tree is None:
        out[prefix + "#n"] = np.zeros(0)
    elif isinstance(tree, str):
        out[prefix + "#s"] = np.array(tree)
    else:
        out[prefix + "#a"] = np.asarray(tree)
    return out


def unpack(flat, prefix=""):
    if prefix + "#d" in flat:
        d = {}
        for ek in flat[prefix + "#d"].tolist():
            key = int(ek[1:]) if ek[0] == "i" else ek[1:]
            d[key] = unpack(flat, prefix + ek + "/")
        return d
    if prefix + "#l" in flat:
        return [unpack(flat, prefix + f"{i}/") for i in range(int(flat[prefix + "#l"]))]
    if prefix + "#n" in flat:
        return None
    if prefix + "#s" in flat:
        return str(flat[prefix + "#s"])
    a = flat[prefix + "#a"]
    return a.item() if a.ndim == 0 else a


def load(path):
    with np.load(path, allow_pickle=False) as z:
        return unpack({k: z[k] for k in z.files})


def save(path, tree):
    np.savez_compressed(path, **pack(tree))
