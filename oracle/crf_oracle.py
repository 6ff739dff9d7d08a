"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

ctypes front-end of ``oracle/crf_oracle.c`` (the CPU restatement of CRFsuite's tagger
arithmetic + GECCO's window/pad/max wrapper + refiner) plus a brute-force
path-enumeration second oracle in pure Python/numpy for tiny cases.

Reference call sites restated: ``gecco/crf/__init__.py:209-258``,
``gecco/_meta.py:124-132``, ``gecco/refine.py:51-64,118-200``; [EXT] CRFsuite 0.12
``crf1d_context.c`` (alpha/beta/marginal/viterbi), ``crf1d_tag.c`` (state score).
"""
import ctypes
import itertools
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcrf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc; a few hundred ms)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "crf_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


class correctly_rounded_exp:
    """``with correctly_rounded_exp(): ...``: the state scores' exp as the correctly rounded function (libquadmath's expq
    rounded to double) instead of libm's -- the checker of the product's reference-bits mode.  Not thread-safe."""

    def __enter__(self):
        lib().oracle_set_exp_mode(1)

    def __exit__(self, *exc):
        lib().oracle_set_exp_mode(0)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_D = ctypes.c_double
_I = ctypes.c_int32
_B = ctypes.c_uint8


def _prep(w, trans, contig_ptr, gene_ptr, attr_id):
    w = np.ascontiguousarray(w, dtype=np.float64)
    trans = np.ascontiguousarray(trans, dtype=np.float64)
    contig_ptr = np.ascontiguousarray(contig_ptr, dtype=np.int32)
    gene_ptr = np.ascontiguousarray(gene_ptr, dtype=np.int32)
    attr_id = np.ascontiguousarray(attr_id, dtype=np.int32)
    if attr_id.size == 0:
        attr_id = np.zeros(1, dtype=np.int32)
    return w, trans, contig_ptr, gene_ptr, attr_id


def windowed_marginals(w, trans, contig_ptr, gene_ptr, attr_id, W, step=1, label=1, pad=True):
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    out = np.zeros(max(n, 1), dtype=np.float64)
    rc = lib().oracle_windowed_marginals(
        _p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), len(contig_ptr) - 1, _p(gene_ptr, _I), _p(attr_id, _I),
        int(W), int(step), int(label), int(bool(pad)), _p(out, _D),
    )
    if rc == -2:
        raise ValueError("invalid window size / step")
    if rc:
        raise RuntimeError(f"oracle_windowed_marginals rc={rc}")
    return out[:n]


def _contig_chunks(contig_ptr, parts):
    """Contiguous runs of contigs with about equal gene counts (for the threaded drivers)."""
    nc = len(contig_ptr) - 1
    parts = max(1, min(int(parts), nc))
    targets = np.linspace(0, float(contig_ptr[-1]), parts + 1)[1:-1]
    cuts = np.unique(np.concatenate(([0], np.searchsorted(contig_ptr, targets), [nc]))).astype(np.int64)
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def windowed_marginals_mt(w, trans, contig_ptr, gene_ptr, attr_id, W, step=1, label=1, pad=True, threads=None, grain=8):
    """Same as `windowed_marginals` on all host cores: OpenMP over ranges of `grain` contigs inside the C library
    (contigs are independent, gecco/crf/__init__.py:244).  Identical output."""
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    out = np.zeros(max(n, 1), dtype=np.float64)
    rc = lib().oracle_windowed_marginals_omp(_p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), len(contig_ptr) - 1,
                                             _p(gene_ptr, _I), _p(attr_id, _I), int(W), int(step), int(label), int(bool(pad)),
                                             _p(out, _D), int(threads or 0), int(grain))
    if rc:
        raise RuntimeError(f"oracle_windowed_marginals_omp rc={rc}")
    return out[:n]


def viterbi_mt(w, trans, contig_ptr, gene_ptr, attr_id, threads=None, grain=8):
    """`viterbi` on all host cores (OpenMP over contig ranges)."""
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    nc = len(contig_ptr) - 1
    lab = np.zeros(max(n, 1), dtype=np.int32)
    sc = np.zeros(max(nc, 1), dtype=np.float64)
    rc = lib().oracle_viterbi_omp(_p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), nc, _p(gene_ptr, _I), _p(attr_id, _I),
                                  _p(lab, _I), _p(sc, _D), int(threads or 0), int(grain))
    if rc:
        raise RuntimeError(f"oracle_viterbi_omp rc={rc}")
    return lab[:n], sc[:nc]


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def full_marginals(w, trans, contig_ptr, gene_ptr, attr_id):
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    nc = len(contig_ptr) - 1
    out = np.zeros((max(n, 1), L), dtype=np.float64)
    ln = np.zeros(max(nc, 1), dtype=np.float64)
    rc = lib().oracle_full_marginals(
        _p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), nc, _p(gene_ptr, _I), _p(attr_id, _I), _p(out, _D), _p(ln, _D)
    )
    if rc:
        raise RuntimeError(f"oracle_full_marginals rc={rc}")
    return out[:n], ln[:nc]


def viterbi(w, trans, contig_ptr, gene_ptr, attr_id):
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    nc = len(contig_ptr) - 1
    lab = np.zeros(max(n, 1), dtype=np.int32)
    sc = np.zeros(max(nc, 1), dtype=np.float64)
    rc = lib().oracle_viterbi(
        _p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), nc, _p(gene_ptr, _I), _p(attr_id, _I), _p(lab, _I), _p(sc, _D)
    )
    if rc:
        raise RuntimeError(f"oracle_viterbi rc={rc}")
    return lab[:n], sc[:nc]


def state_scores(w, gene_ptr, attr_id):
    w = np.ascontiguousarray(w, dtype=np.float64)
    gene_ptr = np.ascontiguousarray(gene_ptr, dtype=np.int32)
    attr_id = np.ascontiguousarray(attr_id, dtype=np.int32)
    if attr_id.size == 0:
        attr_id = np.zeros(1, dtype=np.int32)
    n = len(gene_ptr) - 1
    L = w.shape[1]
    out = np.zeros((max(n, 1), L), dtype=np.float64)
    lib().oracle_state_scores(_p(w, _D), L, _p(gene_ptr, _I), _p(attr_id, _I), n, _p(out, _D))
    return out[:n]


def marginals_seq(state, trans):
    state = np.ascontiguousarray(state, dtype=np.float64)
    trans = np.ascontiguousarray(trans, dtype=np.float64)
    T, L = state.shape
    out = np.zeros((T, L))
    ln = _D(0)
    lib().oracle_marginals_seq(_p(state, _D), _p(trans, _D), T, L, _p(out, _D), ctypes.byref(ln))
    return out, ln.value


def viterbi_seq(state, trans):
    state = np.ascontiguousarray(state, dtype=np.float64)
    trans = np.ascontiguousarray(trans, dtype=np.float64)
    T, L = state.shape
    lab = np.zeros(T, dtype=np.int32)
    sc = _D(0)
    lib().oracle_viterbi_seq(_p(state, _D), _p(trans, _D), T, L, _p(lab, _I), ctypes.byref(sc))
    return lab, sc.value


def viterbi_delta(w, trans, contig_ptr, gene_ptr, attr_id):
    """Labels of the 2-label Viterbi recursion in its difference form, evaluated strictly sequentially: the
    specification the device's label-only kernels reproduce bit for bit (see oracle_viterbi_delta)."""
    w, trans, contig_ptr, gene_ptr, attr_id = _prep(w, trans, contig_ptr, gene_ptr, attr_id)
    A, L = w.shape
    n = int(contig_ptr[-1])
    labels = np.zeros(max(n, 1), dtype=np.int32)
    rc = lib().oracle_viterbi_delta(_p(w, _D), _p(trans, _D), A, L, _p(contig_ptr, _I), len(contig_ptr) - 1,
                                    _p(gene_ptr, _I), _p(attr_id, _I), _p(labels, _I))
    if rc:
        raise ValueError("difference form needs 2 labels and trans[0][1] - trans[1][1] <= trans[0][0] - trans[1][0]")
    return labels[:n]


def segment(p, annotated, contig_ptr, threshold=0.8, n_cds=3, edge_distance=0, trim=True, carry_state=False):
    p = np.ascontiguousarray(p, dtype=np.float64)
    annotated = np.ascontiguousarray(annotated, dtype=np.uint8)
    contig_ptr = np.ascontiguousarray(contig_ptr, dtype=np.int32)
    cap = max(1, len(p))
    seg = np.zeros((cap, 4), dtype=np.int32)
    k = lib().oracle_segment(
        _p(p, _D), _p(annotated, _B), _p(contig_ptr, _I), len(contig_ptr) - 1, _D(threshold), int(n_cds),
        int(edge_distance), int(bool(trim)), int(bool(carry_state)), _p(seg, _I), cap,
    )
    if k < 0:
        raise RuntimeError("oracle_segment overflow")
    return seg[:k].copy()


def segment_antismash(p, annotated, contig_ptr, marker_ptr, marker_id, threshold=0.8, n_cds=5, n_biopfams=5,
                      average_threshold=0.6, trim=True, carry_state=False):
    p = np.ascontiguousarray(p, dtype=np.float64)
    annotated = np.ascontiguousarray(annotated, dtype=np.uint8)
    contig_ptr = np.ascontiguousarray(contig_ptr, dtype=np.int32)
    marker_ptr = np.ascontiguousarray(marker_ptr, dtype=np.int32)
    marker_id = np.ascontiguousarray(marker_id if len(marker_id) else np.zeros(1), dtype=np.int32)
    cap = max(1, len(p))
    seg = np.zeros((cap, 4), dtype=np.int32)
    k = lib().oracle_segment_antismash(
        _p(p, _D), _p(annotated, _B), _p(contig_ptr, _I), len(contig_ptr) - 1, _p(marker_ptr, _I), _p(marker_id, _I),
        _D(threshold), int(n_cds), int(n_biopfams), _D(average_threshold), int(bool(trim)), int(bool(carry_state)),
        _p(seg, _I), cap,
    )
    if k < 0:
        raise RuntimeError("oracle_segment_antismash overflow")
    return seg[:k].copy()


# --------------------------------------------------------------------------------------
# second, independent oracle: brute-force enumeration of all L**T label paths
# --------------------------------------------------------------------------------------
def brute_marginals(state, trans):
    """P(y_t = l) by summing exp(score) over all paths; exact up to fp rounding."""
    state = np.asarray(state, dtype=np.float64)
    trans = np.asarray(trans, dtype=np.float64)
    T, L = state.shape
    scores = []
    paths = list(itertools.product(range(L), repeat=T))
    for y in paths:
        s = sum(state[t, y[t]] for t in range(T)) + sum(trans[y[t - 1], y[t]] for t in range(1, T))
        scores.append(s)
    scores = np.array(scores)
    m = scores.max()
    wts = np.exp(scores - m)
    Z = wts.sum()
    marg = np.zeros((T, L))
    for y, wt in zip(paths, wts):
        for t in range(T):
            marg[t, y[t]] += wt
    return marg / Z, m + np.log(Z)


def brute_viterbi(state, trans):
    """Best path by enumeration; ties -> CRFsuite order is NOT reproduced here, so only
    use on inputs with a unique maximiser."""
    state = np.asarray(state, dtype=np.float64)
    trans = np.asarray(trans, dtype=np.float64)
    T, L = state.shape
    best, besty = -np.inf, None
    for y in itertools.product(range(L), repeat=T):
        s = sum(state[t, y[t]] for t in range(T)) + sum(trans[y[t - 1], y[t]] for t in range(1, T))
        if s > best:
            best, besty = s, y
    return np.array(besty, dtype=np.int32), best
