"""ORACLE / TEST INFRASTRUCTURE ONLY (never imported by gecco_amd).

CPU restatement of ``Cluster.domain_composition`` (/root/reference/gecco/model.py:458-503) as
it is called for the type classifier (/root/reference/gecco/types/__init__.py:118), on packed
arrays.  The reference's arithmetic here *is* numpy (``numpy.sum`` per domain name, then
``composition / (composition.sum() or 1)``), so this file makes the very same numpy calls;
`pairwise_sum` is a second, scalar restatement of numpy's summation order
([EXT] numpy `pairwise_sum`) that the HIP kernel follows, pinned here against ``numpy.sum``.
Parity is not pinned by a reference fixture (the reference holds no expected composition for
a known cluster); it is pinned on numpy itself.
"""
import numpy as np


def domain_composition(names, weights, all_possible=None, normalize=True):
    """model.py:485-503 with `names` / `weights` already extracted (:485-491)."""
    names = np.array(list(names))
    weights = np.array(list(weights), dtype=np.float64)
    unique_names = set(names.tolist())
    if all_possible is None:
        all_possible = np.unique(names)
    composition = np.zeros(len(all_possible))
    for i, dom in enumerate(all_possible):
        if dom in unique_names:
            composition[i] = np.sum(weights[names == dom])
    if normalize:
        return composition / (composition.sum() or 1)
    return composition


def compositions_packed(seg, dom_ptr, dom_col, dom_weight, n_cols, normalize=True):
    """Same, for clusters given as gene ranges over CSR domain rows (column ids instead of names)."""
    seg = np.asarray(seg, dtype=np.int64).reshape(-1, 4)
    out = np.zeros((len(seg), n_cols))
    dom_col = np.asarray(dom_col)
    dom_weight = np.asarray(dom_weight, dtype=np.float64)
    for k, (_, _, a, b) in enumerate(seg):
        r0, r1 = int(dom_ptr[a]), int(dom_ptr[b])
        cols, w = dom_col[r0:r1], dom_weight[r0:r1]
        composition = np.zeros(n_cols)
        for c in sorted(set(int(x) for x in cols if 0 <= x < n_cols)):
            composition[c] = np.sum(w[cols == c])
        out[k] = composition / (composition.sum() or 1) if normalize else composition
    return out


def pairwise_sum(a):
    """numpy's float summation order, scalar by scalar (what ``numpy.sum`` does to a contiguous
    1-D float64 array): the reduction starts at 0.0 and adds `_pw` of every 8192-element
    chunk in turn (numpy's reduction buffer size, `numpy.getbufsize()`)."""
    a = [float(x) for x in a]

    def _pw(lo, n):
        if n < 8:
            res = 0.0
            for i in range(n):
                res += a[lo + i]
            return res
        if n <= 128:
            r = a[lo:lo + 8]
            i = 8
            while i < n - (n % 8):
                for k in range(8):
                    r[k] += a[lo + i + k]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res += a[lo + i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return _pw(lo, n2) + _pw(lo + n2, n - n2)

    res = 0.0
    for lo in range(0, len(a), 8192):
        res += _pw(lo, min(8192, len(a) - lo))
    return res
