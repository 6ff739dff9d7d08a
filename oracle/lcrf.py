"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Independent (numpy/struct) reader for the model file GECCO ships, used to feed
the CPU oracle and to cross-check the product's C++ parser
(`gecco_amd/csrc/crf_model.cpp`).

What it restates:

* `gecco/crf/__init__.py:61-99` (`ClusterCRF.trained`): md5 check of
  ``model.pkl`` against ``model.pkl.md5`` then ``pickle.load``.  The pickle
  names classes of packages that are not installed here, so a stub unpickler
  is used; no code from ``gecco``/``sklearn_crfsuite``/``pycrfsuite`` runs.
* [EXT] CRFsuite 0.12 ``crf1d_model.c`` (dependency ``sklearn-crfsuite ~=0.5.0``
  -> ``python-crfsuite`` -> CRFsuite 0.12; not vendored in /root/reference,
  `pyproject.toml:43`): the on-disk ``lCRF``/``FOMC`` v100 layout --
  48-byte header, ``FEAT`` chunk of 20-byte feature records, two ``CQDB``
  string<->id chunks, ``LFRF``/``AFRF`` reference chunks.

Parity pin: header fields and weights quoted in SURVEY.md §8a row M are
asserted by ``tests/test_oracle_model.py``.
"""
import hashlib
import io
import pickle
import struct
from typing import Dict, List, Tuple

import numpy as np

_ALLOWED = {
    ("gecco.crf", "ClusterCRF"),
    ("sklearn_crfsuite.estimator", "CRF"),
    ("sklearn_crfsuite._fileresource", "FileResource"),
    ("pycrfsuite._logparser", "TrainLogParser"),
}


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return type(name, (_Stub,), {"__module__": module})
        if module == "builtins" and name in {"frozenset", "set", "dict", "list", "tuple"}:
            import builtins

            return getattr(builtins, name)
        raise pickle.UnpicklingError(f"forbidden global {module}.{name}")


def load_pickle(pkl_path: str, md5_path: str = None) -> dict:
    """md5-verify and stub-unpickle ``model.pkl``; returns the ClusterCRF attribute dict
    with ``blob`` = the raw CRFsuite model bytes."""
    data = open(pkl_path, "rb").read()
    if md5_path is not None:
        sig = open(md5_path).read().strip()
        if hashlib.md5(data).hexdigest().upper() != sig.upper():
            raise ValueError("MD5 hash of model data does not match signature")
    obj = _StubUnpickler(io.BytesIO(data)).load()
    st = dict(obj.state)
    crf_state = st["model"].state
    st["crf_params"] = {k: v for k, v in crf_state.items() if k not in ("modelfile", "training_log_")}
    st["blob"] = crf_state["modelfile"].state["__FILE_RESOURCE_DATA__"]
    st["training_log"] = crf_state["training_log_"].state
    return st


def _read_cqdb(blob: bytes, off: int) -> List[str]:
    magic, size, flag, byteorder, bwd_size, bwd_offset = struct.unpack_from("<4sIIIII", blob, off)
    assert magic == b"CQDB", magic
    assert byteorder == 0x62445371
    names = []
    for i in range(bwd_size):
        (rec,) = struct.unpack_from("<I", blob, off + bwd_offset + 4 * i)
        rid, ksize = struct.unpack_from("<II", blob, off + rec)
        assert rid == i
        key = blob[off + rec + 8 : off + rec + 8 + ksize]
        assert key[-1:] == b"\0"
        names.append(key[:-1].decode("utf-8"))
    return names


def parse_lcrf(blob: bytes) -> dict:
    """Parse a CRFsuite ``lCRF`` model into dense tables.

    Returns dict with ``labels`` (list), ``attrs`` (list), ``state`` (A x L f64,
    0 where no feature), ``state_mask`` (A x L bool), ``trans`` (L x L f64, 0 where
    no feature), ``trans_mask``, ``header``.
    """
    hdr = struct.unpack_from("<4sI4sIIIIIIIII", blob, 0)
    (magic, size, typ, version, num_features, L, A, off_feat, off_labels, off_attrs, off_lref, off_aref) = hdr
    assert magic == b"lCRF" and typ == b"FOMC" and version == 100
    assert size == len(blob)
    fmagic, fsize, fnum = struct.unpack_from("<4sII", blob, off_feat)
    assert fmagic == b"FEAT"
    feats = np.frombuffer(
        blob, dtype=np.dtype([("type", "<u4"), ("src", "<u4"), ("dst", "<u4"), ("w", "<f8")]), count=fnum, offset=off_feat + 12
    )
    labels = _read_cqdb(blob, off_labels)
    attrs = _read_cqdb(blob, off_attrs)
    assert len(labels) == L and len(attrs) == A
    state = np.zeros((A, L), dtype=np.float64)
    smask = np.zeros((A, L), dtype=bool)
    trans = np.zeros((L, L), dtype=np.float64)
    tmask = np.zeros((L, L), dtype=bool)
    # walk the reference chunks exactly like the tagger does (crf1dt_state_score /
    # crf1dt_transition_score): attr -> fids -> feature(dst, weight)
    amagic, asize, anum = struct.unpack_from("<4sII", blob, off_aref)
    assert amagic == b"AFRF" and anum == A
    for a in range(A):
        (o,) = struct.unpack_from("<I", blob, off_aref + 12 + 4 * a)
        (n,) = struct.unpack_from("<I", blob, o)
        for fid in struct.unpack_from(f"<{n}I", blob, o + 4):
            f = feats[fid]
            assert f["type"] == 0 and f["src"] == a
            state[a, f["dst"]] += f["w"]
            smask[a, f["dst"]] = True
    lmagic, lsize, lnum = struct.unpack_from("<4sII", blob, off_lref)
    assert lmagic == b"LFRF"
    for i in range(L):
        (o,) = struct.unpack_from("<I", blob, off_lref + 12 + 4 * i)
        (n,) = struct.unpack_from("<I", blob, o)
        for fid in struct.unpack_from(f"<{n}I", blob, o + 4):
            f = feats[fid]
            assert f["type"] == 1 and f["src"] == i
            trans[i, f["dst"]] = f["w"]
            tmask[i, f["dst"]] = True
    return dict(
        header=hdr, labels=labels, attrs=attrs, state=state, state_mask=smask, trans=trans, trans_mask=tmask,
        n_feat=int(fnum), feats=feats,
    )


def load_model(pkl_path: str, md5_path: str = None) -> dict:
    st = load_pickle(pkl_path, md5_path)
    m = parse_lcrf(st["blob"])
    m.update(
        window_size=st["window_size"], window_step=st["window_step"], feature_type=st["feature_type"],
        algorithm=st["algorithm"], significance=st["significance"], significant_features=st["significant_features"],
    )
    m["attr_index"] = {a: i for i, a in enumerate(m["attrs"])}
    return m


def state_features_view(m: dict) -> Dict[Tuple[str, str], float]:
    """[EXT] ``sklearn_crfsuite.CRF.state_features_`` is parsed back from CRFsuite's text
    dump, which prints weights with ``%f`` -- i.e. rounded to 6 decimals."""
    out = {}
    for f in m["feats"]:
        if f["type"] == 0:
            out[(m["attrs"][f["src"]], m["labels"][f["dst"]])] = float("%f" % f["w"])
    return out
