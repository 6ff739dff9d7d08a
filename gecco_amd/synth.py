"""Seeded synthetic workloads for bench.py and the tests.  No reference
data is needed: distributions follow the BGC0001866 fixture's empirical law."""
from typing import Tuple

import numpy as np

SEED = 0x6ECC0
EMBEDDED_TRANS = np.array([[2.669891070463728, -2.599571900486168], [-2.6019205422130995, 2.5683226020688488]])


def synth_model(A: int, rng: np.random.Generator, law: str = "8d") -> Tuple[np.ndarray, np.ndarray]:
    """A x 2 weight table: 58 % of attributes carry an antisymmetric pair (-w, w), 42 % a
    single label; transitions = embedded 2x2.
    law "8d" (the default, the contract's law): SURVEY.md §8d to the letter -- w ~ Laplace(0, 1.7) clipped to [-6.3, 12.7];
    law "genome" (a side point of the bench): w ~ Laplace(-0.4, 1.7) and the Zipf head forced negative -- -0.4 reproduces
    the embedded model's mean w['1']-w['0'] = -0.76, so that most genes lean to label '0' as in real genomes."""
    if law not in ("genome", "8d"):
        raise ValueError(law)
    mag = np.clip(rng.laplace(-0.4 if law == "genome" else 0.0, 1.7, size=A), -6.3, 12.7)
    if law == "genome":
        # the Zipf head (ids < A/50: ubiquitous "housekeeping" domains) argues against clusters
        head = max(1, A // 50)
        mag[:head] = -np.abs(mag[:head])
    both = rng.random(A) < 0.58
    lab = rng.integers(0, 2, size=A)
    w = np.zeros((A, 2))
    w[:, 0] = np.where(both, -mag, np.where(lab == 0, mag, 0.0))
    w[:, 1] = np.where(both, mag, np.where(lab == 1, mag, 0.0))
    return w, EMBEDDED_TRANS.copy()


def synth_contigs(rng: np.random.Generator, lengths, A: int, zipf: float = 1.2, planted: float = 0.0,
                  hot_attrs=None):
    """CSR batch (contig_ptr, gene_ptr, attr_id), int32.  Distinct domains per gene ~
    {0:.30, 1:.35, 2:.17, >=3:.18 as 3+Geometric(.5)}; ids Zipf(1.2) over the A attributes.
    `planted`: fraction of contigs that carry a 10-40 gene run drawn from `hot_attrs`
    (high-weight attributes) so that clusters exist."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = int(lengths.sum())
    u = rng.random(n)
    k = np.where(u < 0.30, 0, np.where(u < 0.65, 1, np.where(u < 0.82, 2, 3)))
    k = np.where(k == 3, 3 + rng.geometric(0.5, size=n) - 1, k).astype(np.int64)
    contig_ptr = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=contig_ptr[1:])
    hot_gene = np.zeros(n, dtype=bool)
    if planted > 0 and hot_attrs is not None and len(hot_attrs):
        for c in np.nonzero(rng.random(len(lengths)) < planted)[0]:
            ln = int(lengths[c])
            run = int(min(ln, rng.integers(10, 41)))
            s = int(contig_ptr[c] + rng.integers(0, ln - run + 1))
            hot_gene[s:s + run] = True
        k = np.where(hot_gene, np.maximum(k, 1), k)
    gene_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(k, out=gene_ptr[1:])
    nnz = int(gene_ptr[-1])
    ranks = np.arange(1, A + 1, dtype=np.float64) ** (-zipf)
    cdf = np.cumsum(ranks / ranks.sum())
    attr = np.minimum(np.searchsorted(cdf, rng.random(nnz)), A - 1).astype(np.int32)
    if hot_gene.any():
        owner = np.repeat(np.arange(n), k)
        hot = hot_gene[owner]
        attr[hot] = rng.choice(np.asarray(hot_attrs, dtype=np.int32), size=int(hot.sum()))
    return contig_ptr.astype(np.int32), gene_ptr.astype(np.int32), attr


def contig_lengths(rng: np.random.Generator, n_contigs: int, total_genes: int = None, median: float = 200.0,
                   sigma: float = 0.5, lo: int = 5, hi: int = 2000) -> np.ndarray:
    ln = np.clip(np.round(rng.lognormal(np.log(median), sigma, size=n_contigs)), lo, hi)
    if total_genes is not None:
        ln = np.clip(np.round(ln * (total_genes / ln.sum())), lo, None)
    return ln.astype(np.int64)


def workload(name: str, seed: int = SEED, law: str = "8d"):
    """Named configurations of BASELINE.json (C2, C3, C5; Cinf = 2e8 genes) -> dict(w, trans, contig_ptr, gene_ptr, attr_id).
    `law`: the weight law of `synth_model` (contig lengths and domain counts do not depend on it; the planted runs
    draw from the law's own 200 most label-leaning attributes)."""
    rng = np.random.default_rng(seed)
    A = 35000
    w, trans = synth_model(A, rng, law="genome")
    if law != "genome":
        w, trans = synth_model(A, np.random.default_rng(seed), law=law)  # (same draws, other law: the rng stream stays aligned)
    hot = np.argsort(w[:, 1] - w[:, 0])[-200:]
    if name == "C2":
        lengths = contig_lengths(rng, 1000)
    elif name == "C3":
        lengths = contig_lengths(rng, 10000, total_genes=2_000_000)
    elif name == "C5":
        lengths = np.full(100, 50000, dtype=np.int64)
    elif name == "Cinf":
        lengths = contig_lengths(rng, 10000, total_genes=2_000_000)
    else:
        raise ValueError(name)
    cptr, gptr, attr = synth_contigs(rng, lengths, A, planted=0.01, hot_attrs=hot)
    if name == "Cinf":  # SURVEY.md 8d "C-infinity": 2e8 genes = 100 concatenated copies of the C3 batch
        reps = 100
        n, nnz = int(cptr[-1]), int(gptr[-1])
        cptr = np.concatenate([[0]] + [cptr[1:].astype(np.int64) + r * n for r in range(reps)]).astype(np.int32)
        gptr = np.concatenate([[0]] + [gptr[1:].astype(np.int64) + r * nnz for r in range(reps)]).astype(np.int32)
        attr = np.tile(attr, reps)
    return dict(name=name, w=w, trans=trans, contig_ptr=cptr, gene_ptr=gptr, attr_id=attr, A=A)
