"""Sharding of a contig batch over ranks / devices.

Contigs are independent on this path (``gecco/crf/__init__.py:244``: one loop iteration per
contig, nothing shared), so a batch shards with no exchange step: every rank scores its own
contigs and the host places the per-gene results back by contig offset.
"""
from typing import List, Sequence, Tuple

import numpy as np


def partition_contigs(contig_lengths: Sequence[int], n_shards: int) -> List[np.ndarray]:
    """Greedy longest-first assignment of contigs to `n_shards` shards, balancing gene counts.
    Returns, per shard, the sorted indices of its contigs (deterministic)."""
    lengths = np.asarray(contig_lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(n_shards, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int64)
    for c in order:
        s = int(np.argmin(load))
        owner[c] = s
        load[s] += lengths[c]
    return [np.nonzero(owner == s)[0] for s in range(n_shards)]


def extract_shard(contig_ptr, gene_ptr, attr_id, contigs: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """CSR sub-batch made of `contigs` (in that order) + the global gene index of each of its genes."""
    contig_ptr = np.asarray(contig_ptr, dtype=np.int64)
    gene_ptr = np.asarray(gene_ptr, dtype=np.int64)
    attr_id = np.asarray(attr_id, dtype=np.int32)
    lens = contig_ptr[contigs + 1] - contig_ptr[contigs]
    sub_cptr = np.zeros(len(contigs) + 1, dtype=np.int64)
    np.cumsum(lens, out=sub_cptr[1:])
    gene_idx = np.concatenate([np.arange(contig_ptr[c], contig_ptr[c + 1]) for c in contigs]) if len(contigs) else np.zeros(0, dtype=np.int64)
    deg = gene_ptr[gene_idx + 1] - gene_ptr[gene_idx]
    sub_gptr = np.zeros(len(gene_idx) + 1, dtype=np.int64)
    np.cumsum(deg, out=sub_gptr[1:])
    if len(gene_idx):
        # gather attribute runs gene by gene (vectorised: repeat starts, add ramps)
        starts = np.repeat(gene_ptr[gene_idx], deg)
        ramp = np.arange(int(sub_gptr[-1])) - np.repeat(sub_gptr[:-1], deg)
        sub_attr = attr_id[starts + ramp]
    else:
        sub_attr = np.zeros(0, dtype=np.int32)
    return sub_cptr.astype(np.int32), sub_gptr.astype(np.int32), sub_attr, gene_idx


def scatter_results(n_genes: int, parts: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    """Inverse of `extract_shard` for per-gene results: parts = [(gene_idx, values), ...]."""
    out = np.full(n_genes, np.nan, dtype=np.float64)
    for gene_idx, values in parts:
        out[gene_idx] = values
    return out
