"""CPU: the composition oracle (numpy restatement of gecco/model.py:458-503) and the scalar
restatement of numpy's summation order that the HIP kernel follows, pinned on numpy.sum."""
import os

import numpy as np

from oracle import composition as oc
from tests.helpers import GOLDEN, read_tsv


def test_pairwise_restatement_is_numpy_sum():
    rng = np.random.default_rng(0)
    for n in list(range(0, 300)) + [1000, 2766, 5000, 8191, 8192, 8193, 10007, 16385, 65536, 100001]:
        a = rng.random(n) * 10.0 ** rng.integers(-8, 8, size=n)
        assert oc.pairwise_sum(a) == float(np.sum(a)), n


def test_domain_composition_by_hand():
    names = ["PF1", "PF2", "PF1", "PF9"]
    w = [0.5, 0.25, 0.125, 1.0]
    comp = oc.domain_composition(names, w, ["PF0", "PF1", "PF2"], normalize=False)
    assert comp.tolist() == [0.0, 0.625, 0.25]
    comp = oc.domain_composition(names, w, ["PF0", "PF1", "PF2"])
    assert comp.tolist() == [0.0, 0.625 / 0.875, 0.25 / 0.875]
    assert oc.domain_composition(names, w, ["PF0"]).tolist() == [0.0]  # `sum() or 1`
    assert oc.domain_composition(names, w).tolist() == (np.array([0.625, 0.25, 1.0]) / 1.875).tolist()  # sorted unique names


def test_packed_equals_named():
    rng = np.random.default_rng(1)
    all_possible = [f"PF{i:05d}" for i in range(60)]
    n_genes = 40
    k = rng.integers(0, 5, size=n_genes)
    dom_ptr = np.concatenate([[0], np.cumsum(k)])
    col = rng.integers(-1, 60, size=int(dom_ptr[-1]))
    w = rng.random(int(dom_ptr[-1]))
    seg = [(0, 1, 0, 7), (0, 2, 9, 9), (0, 3, 10, 40)]
    packed = oc.compositions_packed(seg, dom_ptr, col, w, 60)
    for row, (_, _, a, b) in zip(packed, seg):
        r0, r1 = dom_ptr[a], dom_ptr[b]
        names = [all_possible[c] if c >= 0 else "other" for c in col[r0:r1]]
        assert np.array_equal(row, oc.domain_composition(names, w[r0:r1], all_possible))


def test_golden_cluster_composition(oracle_model):
    """BGC0001866: one cluster of 23 genes, 37 domain rows, 25 distinct domains."""
    feats = read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    names = [r["domain"] for r in feats]
    w = [1 - float(r["pvalue"]) for r in feats]
    all_possible = sorted(oracle_model["attr_index"])
    comp = oc.domain_composition(names, w, all_possible)
    assert abs(comp.sum() - 1.0) <= 1e-15
    assert (comp > 0).sum() == len(set(names) & set(all_possible)) == 25
