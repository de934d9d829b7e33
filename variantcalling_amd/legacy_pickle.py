"""Reading the reference's own pickles without the reference's code.

`--model_file` / `--blacklist` of the reference tools are pickles (dill / pickle) of objects whose classes live in the
un-vendored submodule `ugbio_utils` (`ugbio_filtering.*`; older trees: `ugvc.filtering.*`, `python_pipelines.*`):
`pickle.load` fails on them with ModuleNotFoundError before any data is seen.  What the engine needs from such a file
is data, not behaviour - the scikit-learn estimators per variant-type group, the loci of a blacklist - so the loader
below resolves

  * classes of the libraries the estimators are made of (scikit-learn, numpy, scipy, pandas, builtins) normally,
  * every OTHER global to a plain `Holder` class that records constructor arguments and state and runs nothing,

and `find_estimators` / `find_loci` walk the resulting object graph.  docs/filter_variants_pipeline.md:26-29 (a dict of
named models), docs/howto-callset-filter.md:114,139 (names), SURVEY.md appendix A (the shape: a hierarchical model with
one estimator per group `snp` / `h-indel` / `non-h-indel`) and appendix C (the LFS fixtures this is for:
`exact_gt.model.pkl`, `approximate_gt.model.pkl`, `blacklist_example.chr1_1_1000000.pkl`).  Host logic only.
"""
from __future__ import annotations

import io
import pickle

_REAL_TOPLEVEL = ("sklearn", "numpy", "scipy", "pandas", "builtins", "__builtin__", "collections", "copyreg", "copy_reg",
                  "_codecs", "datetime", "functools", "operator", "xgboost", "joblib")


class Holder:
    """Stand-in for an object of an absent class: keeps what the pickle says about it."""
    _ugvc_origin = ("?", "?")
    _ugvc_args, _ugvc_kwargs, _ugvc_state = (), {}, None          # (protocol >= 2 builds objects without __init__)

    def __init__(self, *args, **kwargs):
        self._ugvc_args = args
        self._ugvc_kwargs = kwargs
        self._ugvc_state = None

    def __setstate__(self, state):
        self._ugvc_state = state
        if isinstance(state, dict):
            self.__dict__.update({k: v for k, v in state.items() if isinstance(k, str) and not k.startswith("_ugvc_")})
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):       # (dict state, slots state)
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update({k: v for k, v in part.items() if isinstance(k, str)})

    def __call__(self, *args, **kwargs):          # a held FUNCTION applied by REDUCE: the call becomes a holder as well
        h = Holder(*args, **kwargs)
        h._ugvc_origin = self._ugvc_origin
        return h

    def __repr__(self):
        return f"<Holder {'.'.join(self._ugvc_origin)}>"


_holders: dict = {}


def _holder_class(module: str, name: str):
    key = (module, name)
    if key not in _holders:
        _holders[key] = type(name, (Holder,), {"_ugvc_origin": key, "__module__": "variantcalling_amd.legacy_pickle"})
    return _holders[key]


# scikit-learn's compiled tree: `Tree.__setstate__` of scikit-learn >= 1.3 refuses the node array of a <= 1.2 pickle
# ("node array from the pickle has an incompatible dtype": no `missing_go_to_left` field) - and the reference pins
# 1.2.2 (setup/environment.yml:399).  Held instead, its state is plain numpy: `model_io.tree_arrays` reads either form.
_HELD_ON_RETRY = {("sklearn.tree._tree", "Tree")}


class _ShimUnpickler(pickle.Unpickler):
    hold_trees = False

    def find_class(self, module, name):
        top = module.split(".", 1)[0]
        if self.hold_trees and (module, name) in _HELD_ON_RETRY:
            return _holder_class(module, name)
        if top in _REAL_TOPLEVEL:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                pass                                  # a class a newer / older library no longer has: hold it
        return _holder_class(module, name)


def is_tree_state_mismatch(exc: BaseException) -> bool:
    """scikit-learn's `Tree.__setstate__` refusing the node / value arrays of another generation's pickle (the messages of
    sklearn/tree/_tree.pyx: "node array from the pickle has an incompatible dtype", "... value array ...", "Wrong dimensions
    for node array from the pickle"; scikit-learn < 1.3: "Did not recognise loaded array layout" / "... dimensions"; a pickle from
    another platform: Cython's "Buffer dtype mismatch, expected ..." raised from the same __setstate__)."""
    if not isinstance(exc, ValueError):
        return False
    msg = str(exc)
    if "from the pickle" in msg and ("array" in msg or "n_classes" in msg):
        return True
    return "Did not recognise loaded array" in msg or "Buffer dtype mismatch" in msg


def load(path_or_bytes):
    """The object graph of a pickle, absent classes as Holders."""
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    try:
        return _ShimUnpickler(io.BytesIO(raw)).load()
    except ValueError as exc:
        # an estimator pickled by another scikit-learn generation: keep the compiled trees as data (see above).  Any OTHER
        # ValueError (a corrupt or unrelated pickle) is the caller's to see as it is, not after a second unpickling.
        if not is_tree_state_mismatch(exc):
            raise
        up = _ShimUnpickler(io.BytesIO(raw))
        up.hold_trees = True
        return up.load()


def _is_estimator(o) -> bool:
    return hasattr(o, "tree_") or hasattr(o, "estimators_") or (hasattr(o, "get_booster") and hasattr(o, "predict"))


def _children(o):
    if isinstance(o, dict):
        return list(o.items())
    if isinstance(o, (list, tuple)):
        return list(enumerate(o))
    if isinstance(o, Holder):
        kids = [(k, v) for k, v in vars(o).items() if not k.startswith("_ugvc_")]
        kids += [(f"arg{i}", a) for i, a in enumerate(o._ugvc_args)]
        if o._ugvc_state is not None and not isinstance(o._ugvc_state, dict):
            kids.append(("state", o._ugvc_state))
        return kids
    return []


def find_estimators(obj, group_names=("snp", "h-indel", "non-h-indel"), max_depth: int = 8):
    """{model name: [estimator or None per group]} from a loaded model file: a dict of named models whose values hold,
    somewhere below, one estimator per variant-type group keyed by the group's name (any key containing it), or one
    estimator for all groups."""
    def per_group(model, depth=0):
        if _is_estimator(model):
            return [model] * len(group_names)
        found = [None] * len(group_names)
        lone = []

        def walk(o, key, d):
            if d > max_depth:
                return
            if _is_estimator(o):
                ks = str(key).lower()
                hits = [g for g, gn in enumerate(group_names) if ks == gn or gn in ks.replace("_", "-")]
                # "h-indel" is a substring of "non-h-indel": the longest matching name wins
                if hits:
                    g = max(hits, key=lambda g: len(group_names[g]))
                    if found[g] is None:
                        found[g] = o
                else:
                    lone.append(o)
                return
            for k, v in _children(o):
                walk(v, k if not isinstance(k, int) else key, d + 1)
        walk(model, "", 0)
        if all(f is None for f in found) and lone:
            return [lone[0]] * len(group_names) if len(lone) == 1 else (lone + [None] * len(group_names))[: len(group_names)]
        return found

    if not isinstance(obj, dict):
        obj = {"model": obj}
    out = {}
    for name, model in obj.items():
        groups = per_group(model)
        if any(g is not None for g in groups):
            out[str(name)] = groups
    return out


def find_loci(obj, max_depth: int = 8):
    """(chrom, pos) pairs anywhere in a loaded blacklist pickle: pandas frames / indexes with chrom and pos, tuples."""
    loci = []

    def walk(o, d):
        if d > max_depth:
            return
        cols = getattr(o, "columns", None)
        if cols is not None and "chrom" in list(cols) and "pos" in list(cols):
            loci.extend(zip(o["chrom"], o["pos"]))
            return
        idx = getattr(o, "index", None)
        if idx is not None and hasattr(idx, "names") and list(idx.names)[:2] == ["chrom", "pos"]:
            loci.extend((t[0], t[1]) for t in idx.tolist())
            return
        if hasattr(o, "names") and hasattr(o, "tolist") and list(getattr(o, "names", []))[:2] == ["chrom", "pos"]:
            loci.extend((t[0], t[1]) for t in o.tolist())
            return
        if isinstance(o, (list, tuple, set, frozenset)) and o and all(
                isinstance(v, tuple) and len(v) >= 2 and isinstance(v[0], str) and isinstance(v[1], (int,)) for v in list(o)[:8]):
            loci.extend((v[0], int(v[1])) for v in o)
            return
        for _, v in _children(o):
            walk(v, d + 1)
        if isinstance(o, (set, frozenset)):
            for v in o:
                walk(v, d + 1)
    walk(obj, 0)
    return loci
