"""Weighted domain compositions of called clusters, computed on the device.

Mirrors ``Cluster.domain_composition`` (``/root/reference/gecco/model.py:458-503``) and the way
``TypeClassifier.predict_types`` assembles its input matrix
(``/root/reference/gecco/types/__init__.py:118``:
``numpy.array([c.domain_composition(self.model.attributes_) for c in clusters])``).
Names, argument meaning and defaults follow the reference; the sums and the normalisation run in
``gecco_crf_domain_composition`` (bit-identical to numpy's summation order), the host side only
turns names into column ids and p-values into weights exactly as the reference does
(``1 - v`` or ``-log10(v)``).  The random forest itself is out of scope.
"""
import math
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _native


def _columns(all_possible: Sequence[str]):
    first, dups = {}, []
    for i, name in enumerate(all_possible):
        if name in first:
            dups.append((i, first[name]))  # the reference fills every occurrence (model.py:497-499)
        else:
            first[name] = i
    return first, dups


def _weight(domain, minlog_weights: bool, pvalue: bool) -> float:
    v = domain.pvalue if pvalue else domain.i_evalue
    return -math.log10(v) if minlog_weights else 1 - v


def cluster_compositions(clusters: Iterable, all_possible: Sequence[str], normalize: bool = True,
                         minlog_weights: bool = False, pvalue: bool = True, device: int = 0) -> np.ndarray:
    """``numpy.array([c.domain_composition(all_possible, normalize, minlog_weights, pvalue) for c in clusters])``."""
    all_possible = list(all_possible)
    col_of, dups = _columns(all_possible)
    seg, dom_ptr, dom_col, dom_w = [], [0], [], []
    n_genes = 0
    for k, cluster in enumerate(clusters):
        a = n_genes
        for gene in cluster.genes:
            for d in gene.protein.domains:
                dom_col.append(col_of.get(d.name, -1))
                dom_w.append(_weight(d, minlog_weights, pvalue))
            dom_ptr.append(len(dom_col))
            n_genes += 1
        seg.append((0, k + 1, a, n_genes))
    out = _native.domain_composition(np.array(seg, dtype=np.int32).reshape(-1, 4), dom_ptr, dom_col, dom_w, len(all_possible),
                                     normalize=bool(normalize) and not dups, device=device)
    if dups:  # duplicated names in `all_possible`: copy the column, then normalise like numpy does
        for i, j in dups:
            out[:, i] = out[:, j]
        if normalize:
            for r in range(out.shape[0]):
                out[r] = out[r] / (out[r].sum() or 1)
    return out


def domain_composition(cluster, all_possible: Optional[Sequence[str]] = None, normalize: bool = True,
                       minlog_weights: bool = False, pvalue: bool = True, device: int = 0) -> np.ndarray:
    """Drop-in for ``Cluster.domain_composition`` (same arguments; ``all_possible=None`` means the
    sorted distinct names of the cluster itself, model.py:494-495)."""
    if all_possible is None:
        all_possible = sorted({d.name for g in cluster.genes for d in g.protein.domains})
    return cluster_compositions([cluster], all_possible, normalize, minlog_weights, pvalue, device)[0]


def table_compositions(seg: np.ndarray, order: List[str], feature_protein_id: Sequence[str], feature_domain: Sequence[str],
                       feature_pvalue: Sequence[float], feature_domain_start: Sequence[int], all_possible: Sequence[str],
                       normalize: bool = True, device: int = 0) -> np.ndarray:
    """Columnar variant for ``gecco_amd.predict``: `seg` rows from ``gecco_crf_segment`` over the genes
    listed in `order`; the domain rows of a gene are its feature-table rows (model.py:688-706)
    stably sorted by domain start, which is the order ``predict_probabilities`` leaves
    ``protein.domains`` in (crf/__init__.py:200-201)."""
    all_possible = list(all_possible)
    col_of, dups = _columns(all_possible)
    if dups:
        raise ValueError("duplicated names in `all_possible`")
    gene_of = {pid: i for i, pid in enumerate(order)}
    rows_of: List[List[int]] = [[] for _ in order]
    for r, pid in enumerate(feature_protein_id):
        i = gene_of.get(pid)
        if i is not None:
            rows_of[i].append(r)
    dom_ptr, dom_col, dom_w = [0], [], []
    for rows in rows_of:
        for r in sorted(rows, key=lambda r: feature_domain_start[r]):
            dom_col.append(col_of.get(feature_domain[r], -1))
            dom_w.append(1 - feature_pvalue[r])
        dom_ptr.append(len(dom_col))
    return _native.domain_composition(seg, dom_ptr, dom_col, dom_w, len(all_possible), normalize=normalize, device=device)


def packed_compositions(seg: np.ndarray, packed, feature_domain: Sequence[str], feature_pvalue: Sequence[float],
                        all_possible: Sequence[str], normalize: bool = True, device: int = 0) -> np.ndarray:
    """`table_compositions` on the row ordering ``packing.pack_columns`` already computed
    (`packed.row_order` / `packed.row_ptr`: feature rows by gene position and domain start): no
    per-row Python work."""
    all_possible = list(all_possible)
    col_of, dups = _columns(all_possible)
    if dups:
        raise ValueError("duplicated names in `all_possible`")
    dom = np.asarray(feature_domain, dtype=object)[packed.row_order]
    names, inverse = np.unique(dom.astype(str), return_inverse=True) if len(dom) else (np.zeros(0, dtype=str), np.zeros(0, dtype=np.int64))
    col_of_name = np.fromiter((col_of.get(str(nm), -1) for nm in names), dtype=np.int32, count=len(names))
    dom_col = col_of_name[inverse] if len(dom) else np.zeros(0, dtype=np.int32)
    dom_w = 1 - np.asarray(feature_pvalue, dtype=np.float64)[packed.row_order]
    return _native.domain_composition(seg, packed.row_ptr.astype(np.int32), dom_col, dom_w, len(all_possible),
                                      normalize=normalize, device=device)
