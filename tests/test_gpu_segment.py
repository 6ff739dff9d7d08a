"""GPU parity of row R on packed arrays: gecco_crf_segment vs the oracle's restatement of
GeneGrouper / ClusterRefiner (gecco/refine.py:51-64,118-200)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_case(rng, n_contigs, max_len):
    p_all, ann_all, cptr = [], [], [0]
    for c in range(n_contigs):
        n = int(rng.integers(0, max_len))
        mode = rng.random()
        p = np.clip(rng.normal(0.9 if mode < 0.4 else 0.3, 0.3, size=n), 0, 1)
        p[rng.random(n) < 0.08] = np.nan
        if rng.random() < 0.1:
            p[:] = np.nan  # a contig without any prediction inherits the grouper state
        p_all.append(p)
        ann_all.append(rng.random(n) < 0.65)
        cptr.append(cptr[-1] + n)
    return np.concatenate(p_all), np.concatenate(ann_all).astype(np.uint8), np.array(cptr, dtype=np.int32)


@pytest.mark.parametrize("n_cds,edge,trim", [(3, 0, True), (1, 0, False), (2, 2, True), (5, 1, True), (1, 10, True)])
def test_segment_matches_oracle(n_cds, edge, trim):
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(100 * n_cds + edge)
    for n_contigs, max_len in [(1, 50), (40, 80), (3000, 60), (5, 5000), (3, 40000)]:
        p, ann, cptr = _random_case(rng, n_contigs, max_len)
        for carry in (False, True):  # one grouper per contig (the CLI) / one for the whole call
            exp = orc.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            got = nat.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            assert got.tolist() == exp.tolist()


def test_segment_golden(oracle_model):
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc
    from tests.helpers import golden_csr

    ids, cptr, gptr, attr, expected, ann = golden_csr(oracle_model["attr_index"])
    seg = nat.segment(expected, ann, cptr, 0.8, 3, 0, True)
    assert seg.tolist() == [[0, 1, 0, 23]]
    assert nat.segment(np.zeros(0), np.zeros(0, dtype=np.uint8), [0]).shape == (0, 4)


def test_segment_long_runs_and_empty_contigs():
    """Runs that span many scan blocks, contigs without genes, runs without any annotated gene."""
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(5)
    n = 30000
    p = np.full(n, 0.95)
    p[rng.integers(0, n, size=6)] = 0.1          # a handful of cuts: runs of thousands of genes
    p[20000:20100] = np.nan
    ann = (rng.random(n) < 0.5).astype(np.uint8)
    ann[:3000] = 0                                # the first run trims to nothing
    cptr = np.array([0, 0, 12000, 12000, 12001, n, n], dtype=np.int32)
    for n_cds, edge, trim in [(3, 0, True), (1, 5, True), (0, 0, True), (2, 0, False), (3, 100000, True)]:
        for carry in (False, True):
            exp = orc.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            got = nat.segment(p, ann, cptr, 0.8, n_cds, edge, trim, carry_state=carry)
            assert got.tolist() == exp.tolist(), (n_cds, edge, trim, carry)


def _random_markers(rng, n, n_markers=130, rate=0.25):
    """CSR of marker-domain indices per gene (what the packer derives from the domain names)."""
    cnt = np.where(rng.random(n) < rate, rng.integers(1, 4, size=n), 0)
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    # few distinct markers per neighbourhood, so that the distinct count hovers around the criterion
    ids = (rng.integers(0, 12, size=int(ptr[-1])) + 7 * (np.repeat(np.arange(n), cnt) // 50)) % n_markers
    return ptr, ids.astype(np.int32)


@pytest.mark.parametrize("n_cds,n_bio,avg,trim", [(5, 5, 0.6, True), (1, 1, 0.0, False), (3, 2, 0.9, True), (2, 0, 0.85, True)])
def test_segment_antismash_matches_oracle(n_cds, n_bio, avg, trim):
    """criterion "antismash" (refine.py:157-163) on the device: mean probability, distinct marker domains, genes."""
    from gecco_amd import _native as nat
    from oracle import crf_oracle as orc

    rng = np.random.default_rng(7 + n_cds)
    seen = 0
    for n_contigs, max_len in [(1, 50), (40, 80), (3000, 60), (5, 5000)]:
        p, ann, cptr = _random_case(rng, n_contigs, max_len)
        mptr, mid = _random_markers(rng, len(p))
        for carry in (False, True):
            exp = orc.segment_antismash(p, ann, cptr, mptr, mid, 0.8, n_cds, n_bio, avg, trim, carry_state=carry)
            got = nat.segment(p, ann, cptr, 0.8, n_cds, 0, trim, carry_state=carry, criterion="antismash", n_biopfams=n_bio,
                              average_threshold=avg, marker_ptr=mptr, marker_id=mid)
            assert got.tolist() == exp.tolist()
            seen += len(exp)
    assert seen > 20
    with pytest.raises(ValueError, match="Unknown cluster filtering criterion"):
        nat.segment(p, ann, cptr, criterion="nonsense")
    with pytest.raises(Exception, match="marker"):
        nat.segment(p, ann, cptr, criterion="antismash")


def test_segment_antismash_equals_the_object_refiner():
    """Packed arrays against gecco_amd.refine.ClusterRefiner(criterion="antismash") on objects (the mirror of
    refine.py:118-200): the same clusters when no mean lies within rounding of the threshold."""
    from gecco_amd import _native as nat
    from gecco_amd import refine
    from gecco_amd.model import Cluster, Domain, Gene, Protein, Source, Strand

    rng = np.random.default_rng(3)
    markers = sorted(refine.BIO_PFAMS)
    others = [f"PF{90000 + i}" for i in range(40)]
    genes, p_all, ann, mptr, mid, cptr = [], [], [], [0], [], [0]
    for c in range(30):
        src = Source(f"contig{c:02d}")
        n = int(rng.integers(5, 120))
        base = 0.9 if rng.random() < 0.5 else 0.4
        for g in range(n):
            names = []
            for _ in range(int(rng.integers(0, 4))):
                names.append(str(rng.choice(markers[:15])) if rng.random() < 0.5 else str(rng.choice(others)))
            p = float(np.clip(rng.normal(base, 0.25), 0, 1))
            doms = [Domain(nm, i, i + 1, "Pfam", 0.0, 0.0, p) for i, nm in enumerate(names)]
            genes.append(Gene(src, 10 * g, 10 * g + 9, Strand.Coding, Protein(f"{src.id}_{g}", None, doms), _probability=p))
            p_all.append(p)
            ann.append(1 if doms else 0)
            ms = sorted({markers.index(nm) for nm in names if nm in refine.BIO_PFAMS})
            mid.extend(ms)
            mptr.append(len(mid))
        cptr.append(len(p_all))
    for kw in (dict(n_cds=5, n_biopfams=5, average_threshold=0.6), dict(n_cds=2, n_biopfams=2, average_threshold=0.85),
               dict(n_cds=3, n_biopfams=1, average_threshold=0.5, trim=False)):
        ref = refine.ClusterRefiner(criterion="antismash", threshold=0.8, cluster_type=Cluster, **kw)
        exp = []
        import itertools

        for ci, (_, group) in enumerate(itertools.groupby(genes, key=lambda g: g.source.id)):
            for cl in ref.iter_clusters(list(group)):
                idx = [i for i in range(cptr[ci], cptr[ci + 1]) if genes[i] is cl.genes[0]][0]
                exp.append([ci, int(cl.id.rsplit("_", 1)[1]), idx, idx + len(cl.genes)])
        got = nat.segment(p_all, ann, cptr, 0.8, kw["n_cds"], 0, kw.get("trim", True), criterion="antismash",
                          n_biopfams=kw["n_biopfams"], average_threshold=kw["average_threshold"], marker_ptr=mptr, marker_id=mid)
        assert got.tolist() == exp
    assert len(exp) > 3
