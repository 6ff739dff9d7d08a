"""Randomised sweep of the drop-in class on `Gene` objects: contigs in or out of (source.id, start) order, domain lists in
or out of start order, repeated and unknown domains, empty genes, contigs shorter than the window with and without padding,
protein and domain features -- against the Python statements of the reference's loop (gecco/crf/__init__.py:199-273, restated
here with the oracle for the arithmetic): the genes come back in the reference's order with the oracle's probabilities (bit
for bit: the class runs in reference-bits mode), fresh objects, domains annotated with the model's cluster weights."""
import operator
import warnings

import numpy as np
import pytest

import torch  # noqa: F401  (before libgecco_crf.so: the wheel's own HIP runtime has to be the first one loaded)

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(60))
def test_predict_probabilities_on_random_objects(oracle_model, seed):
    from gecco_amd.crf import ClusterCRF
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(7000 + seed)
    crf = ClusterCRF.trained(GOLDEN)
    crf.feature_type = "domain" if rng.random() < 0.25 else "protein"
    pad = bool(rng.random() < 0.7)
    attrs = oracle_model["attrs"]
    index = {a: i for i, a in enumerate(attrs)}
    in_order = bool(rng.random() < 0.6)
    shared_source = bool(rng.random() < 0.5)
    genes = []
    for c in range(int(rng.integers(1, 14))):
        src = Source(f"ctg{c:03d}")
        n = int(rng.choice([1, 3, 19, 20, 21, 60, 250])) if rng.random() < 0.6 else int(rng.integers(1, 90))
        for i in range(n):
            k = int(rng.integers(0, 5))
            doms = []
            for j in range(k):
                name = attrs[int(rng.integers(0, min(len(attrs), 60)))] if rng.random() < 0.9 else f"PFX{int(rng.integers(0, 3))}"
                st = int(rng.integers(0, 300))
                doms.append(Domain(name, st, st + 20, "Pfam", 1e-10, 1e-12))
            if in_order and rng.random() < 0.7:
                doms.sort(key=operator.attrgetter("start"))
            genes.append(Gene(src if shared_source else Source(f"ctg{c:03d}"), 100 * i, 100 * i + 90, Strand.Coding,
                              Protein(f"c{c:03d}_{i:04d}", None, doms)))
    if not in_order:
        rng.shuffle(genes)

    # the reference's statements, with the oracle for the arithmetic
    ref = sorted(genes, key=operator.attrgetter("source.id", "start"))
    ref_dom_order = [[d.name for d in sorted(g.protein.domains, key=operator.attrgetter("start"))] for g in ref]
    cptr, gptr, attr, owner = [0], [0], [], []
    prev = None
    for gi, (g, names) in enumerate(zip(ref, ref_dom_order)):
        if prev is not None and g.source.id != prev:
            cptr.append(len(gptr) - 1)
        prev = g.source.id
        if crf.feature_type == "protein":
            seen = []
            for nm in names:
                if nm not in seen:
                    seen.append(nm)
            attr += [index[nm] for nm in seen if nm in index]
            gptr.append(len(attr))
            owner.append(gi)
        else:
            for nm in names or [None]:
                if nm is not None and nm in index:
                    attr.append(index[nm])
                gptr.append(len(attr))
                owner.append(gi)
    cptr.append(len(gptr) - 1)
    cptr, gptr, attr = np.array(cptr, np.int32), np.array(gptr, np.int32), np.array(attr, np.int32)
    with orc.correctly_rounded_exp():
        ep = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, pad)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = crf.predict_probabilities(genes, pad=pad)
    case = dict(seed=seed, feature_type=crf.feature_type, pad=pad, in_order=in_order, n=len(genes))
    assert [g.protein.id for g in out] == [g.protein.id for g in ref], case
    assert all(a is not b for a, b in zip(out, ref)), case
    w1 = crf.model.cluster_weights_
    owner = np.array(owner)
    for gi, (g, names) in enumerate(zip(out, ref_dom_order)):
        items = ep[owner == gi]
        assert [d.name for d in g.protein.domains] == names, case  # (sorted by start, like the caller's own lists now)
        skipped = bool(np.isnan(items).any())
        if crf.feature_type == "protein":
            want = None if skipped else float(items[0])
            assert g._probability == want or (want is None and g._probability is None), (case, gi)
            for d in g.protein.domains:
                assert d.probability == want, (case, gi)
        elif not skipped:
            if names:
                assert [d.probability for d in g.protein.domains] == [float(x) for x in items], (case, gi)
            else:
                assert g._probability == float(items[0]), (case, gi)
        for d in g.protein.domains:
            assert d.cluster_weight == w1.get(d.name), (case, gi)
    # the caller's domain lists were sorted in place (crf/__init__.py:200-201)
    for g in genes:
        starts = [d.start for d in g.protein.domains]
        assert starts == sorted(starts), case
