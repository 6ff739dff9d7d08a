"""Shared test helpers: golden-table readers and seeded synthetic workloads."""
import csv
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_tsv(path):
    with open(path) as fh:
        return list(csv.DictReader(fh, delimiter="\t"))


def golden_csr(attr_index):
    """BGC0001866 fixture -> (protein ids, contig_ptr, gene_ptr, attr_id, expected p, annotated).

    Genes in genes.tsv order (already sorted by start); per gene the *set* of domains in
    first-occurrence order of features.tsv rows sorted by domain_start
    (gecco/crf/__init__.py:200-201, features.py:31-35)."""
    genes = read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    feats = read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    by_gene = {}
    for r in feats:
        by_gene.setdefault(r["protein_id"], []).append(r)
    ids, gptr, attrs, exp, ann = [], [0], [], [], []
    for g in genes:
        rows = sorted(by_gene.get(g["protein_id"], []), key=lambda r: int(r["domain_start"]))
        seen = []
        for r in rows:
            if r["domain"] not in seen:
                seen.append(r["domain"])
        for d in seen:
            if d in attr_index:
                attrs.append(attr_index[d])
        gptr.append(len(attrs))
        ids.append(g["protein_id"])
        exp.append(float(g["average_p"]))
        ann.append(1 if rows else 0)
    return (
        ids,
        np.array([0, len(ids)], dtype=np.int32),
        np.array(gptr, dtype=np.int32),
        np.array(attrs, dtype=np.int32),
        np.array(exp),
        np.array(ann, dtype=np.uint8),
    )


def synth_model(A, rng, L=2):
    """Synthetic weight table per SURVEY.md §8d (C2): 58 % of attrs carry an antisymmetric
    pair (w,-w), 42 % a single label; w ~ Laplace(0,1.7) clipped to [-6.3, 12.7];
    transitions = the embedded model's 2x2."""
    w = np.zeros((A, L))
    mag = np.clip(rng.laplace(0.0, 1.7, size=A), -6.3, 12.7)
    both = rng.random(A) < 0.58
    lab = rng.integers(0, L, size=A)
    for y in range(L):
        w[:, y] = np.where(both, mag if y == 1 else -mag, np.where(lab == y, mag, 0.0)) if L == 2 else 0
    if L != 2:
        w = np.clip(rng.laplace(0.0, 1.7, size=(A, L)), -6.3, 12.7)
    trans = np.array([[2.669891070463728, -2.599571900486168], [-2.6019205422130995, 2.5683226020688488]])
    if L != 2:
        trans = rng.normal(0, 1.5, size=(L, L))
    return w, trans


def synth_contigs(rng, lengths, A, zipf=1.2):
    """CSR batch: per gene #distinct domains ~ {0:.30, 1:.35, 2:.17, >=3:.18 as 3+Geom(.5)},
    ids Zipf(1.2) over A attrs."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = int(lengths.sum())
    u = rng.random(n)
    k = np.where(u < 0.30, 0, np.where(u < 0.65, 1, np.where(u < 0.82, 2, 3)))
    extra = rng.geometric(0.5, size=n) - 1
    k = np.where(k == 3, 3 + extra, k).astype(np.int64)
    gene_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(k, out=gene_ptr[1:])
    nnz = int(gene_ptr[-1])
    ranks = np.arange(1, A + 1, dtype=np.float64) ** (-zipf)
    cdf = np.cumsum(ranks / ranks.sum())
    attr = np.searchsorted(cdf, rng.random(nnz)).astype(np.int32)
    attr = np.minimum(attr, A - 1)
    contig_ptr = np.zeros(len(lengths) + 1, dtype=np.int32)
    np.cumsum(lengths, out=contig_ptr[1:])
    return contig_ptr, gene_ptr.astype(np.int32), attr


# ---- vectors produced by the reference's own Python (tools/gen_reference_fixtures.py -> tests/golden/ref_*.json.gz) ----------
def load_ref(name):
    import gzip
    import json

    with gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rt") as fh:
        return json.load(fh)["cases"]


def genes_from_crf_case(case):
    """`gecco_amd.model` objects for a case of ref_predict_probabilities, in the INPUT order the reference was given
    (rows: contig id, protein id, start, end, strand, [[domain, start, end], ...]; hmm / e-values constant)."""
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    sources, genes = {}, []
    for cid, pid, start, end, strand, doms in case["genes"]:
        src = sources.setdefault(cid, Source(cid))
        genes.append(Gene(src, start, end, Strand(strand),
                          Protein(pid, None, [Domain(name, ds, de, "Pfam", 1e-5, 1e-7) for name, ds, de in doms])))
    return genes


def genes_from_refiner_case(case):
    """rows: contig id, protein id, start, end, probability or None, [domain names]."""
    from gecco_amd.model import Domain, Gene, Protein, Source, Strand

    sources, genes = {}, []
    for cid, pid, start, end, prob, doms in case["genes"]:
        src = sources.setdefault(cid, Source(cid))
        genes.append(Gene(src, start, end, Strand.Coding,
                          Protein(pid, None, [Domain(name, 1, 51, "Pfam", 1e-5, 1e-7, probability=prob) for name in doms]),
                          _probability=prob))
    return genes


def pack_refiner_case(case, markers=None):
    """The packed arrays `gecco_crf_segment` / the oracle take for a refiner case: genes by (contig id, start, end) -- a stable
    sort, like the reference's two sorts (refine.py:189-192) --, NaN for a missing probability; returns
    (ids, contig ids, p, annotated, contig_ptr, marker_ptr, marker_id)."""
    rows = sorted(case["genes"], key=lambda r: r[0])            # sorted(genes, key=source.id): stable
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r)
    ids, cids, p, ann, cptr, mptr, mid = [], [], [], [], [0], [0], []
    mindex = {m: i for i, m in enumerate(markers or [])}
    for cid in sorted(by):
        seq = sorted(by[cid], key=lambda r: (r[2], r[3]))        # (start, end): stable
        cids.append(cid)
        for r in seq:
            ids.append(r[1])
            p.append(float("nan") if r[4] is None else r[4])
            ann.append(1 if r[5] else 0)
            mid.extend(sorted({mindex[d] for d in r[5] if d in mindex}))
            mptr.append(len(mid))
        cptr.append(len(ids))
    return (ids, cids, np.array(p, dtype=np.float64), np.array(ann, dtype=np.uint8), np.array(cptr, dtype=np.int32),
            np.array(mptr, dtype=np.int32), np.array(mid, dtype=np.int32))
