"""Cluster calling behind the CRF: threshold run-length segmentation of per-gene
probabilities, edge trimming and validation.

Mirror of ``/root/reference/gecco/refine.py`` (``GeneGrouper`` ``:51-64``, ``ClusterRefiner``
``:68-200``) with the same constructor arguments, defaults and semantics:

* a gene is "in" when ``average_probability > threshold`` (strict); a gene WITHOUT a
  probability inherits the state of the previous gene -- and the grouper is created once per
  ``iter_clusters`` call, so that state also carries over from one contig to the next;
* every maximal run of "in" genes of a contig (genes sorted by ``(start, end)``) becomes
  ``Cluster(f"{contig}_cluster_{i}")``, numbered from 1 *before* filtering;
* ``trim`` drops genes without domains from both ends;
* criterion ``gecco``: ``#annotated genes >= n_cds`` and ``#(cluster genes - edge genes) >=
  n_cds`` where edge genes are the first/last ``edge_distance`` annotated genes of the contig;
  criterion ``antismash``: mean p >= ``average_threshold``, >= ``n_biopfams`` distinct
  biosynthetic Pfams, >= ``n_cds`` genes.

The segmentation itself is also available on packed arrays -- on the device through
``gecco_crf_segment`` (include/gecco_crf.h) and, for this object-level class, through the same
few lines of host logic below (it is integer bookkeeping on a handful of genes per cluster).
"""
import itertools

import numpy
import operator
import os
from typing import Any, Iterator, List, Optional, Sequence, Tuple

__all__ = ["BIO_PFAMS", "ClusterRefiner", "segment_runs"]

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bio_pfams.txt")) as _fh:
    BIO_PFAMS = frozenset(line.strip() for line in _fh if line.strip() and not line.startswith("#"))


def _cluster_class():
    try:
        from gecco.model import Cluster  # type: ignore

        return Cluster
    except Exception:
        from .model import Cluster

        return Cluster


def segment_runs(probabilities: Sequence[Optional[float]], threshold: float, state: bool = False) -> Tuple[List[Tuple[int, int]], bool]:
    """Maximal runs [a, b) of "in" positions; ``None`` inherits the running state.  Returns the
    runs and the final state (to be threaded into the next contig, as the reference does)."""
    runs: List[Tuple[int, int]] = []
    begin = -1
    for i, p in enumerate(probabilities):
        if p is not None:
            state = p > threshold
        if state and begin < 0:
            begin = i
        elif not state and begin >= 0:
            runs.append((begin, i))
            begin = -1
    if begin >= 0:
        runs.append((begin, len(probabilities)))
    return runs, state


class ClusterRefiner:
    """Post-processor extracting contiguous clusters from CRF predictions."""

    def __init__(self, *, threshold: float = 0.8, criterion: str = "gecco", n_cds: int = 5, n_biopfams: int = 5,
                 average_threshold: float = 0.6, edge_distance: int = 0, trim: bool = True,
                 cluster_type: Any = None) -> None:
        self.threshold = threshold
        self.criterion = criterion
        self.n_cds = n_cds
        self.n_biopfams = n_biopfams
        self.average_threshold = average_threshold
        self.edge_distance = edge_distance
        self.trim = trim
        self._cluster_type = cluster_type

    def iter_clusters(self, genes: List[Any]) -> Iterator[Any]:
        Cluster = self._cluster_type or _cluster_class()
        if self.criterion not in ("gecco", "antismash"):
            raise ValueError(f"Unknown cluster filtering criterion: {self.criterion}")
        key = operator.attrgetter("source.id")
        state = False  # one grouper per call: its state spans contigs (refine.py:186)
        for seq_id, group in itertools.groupby(sorted(genes, key=key), key=key):
            seq = sorted(group, key=operator.attrgetter("start", "end"))
            runs, state = segment_runs([g.average_probability for g in seq], self.threshold, state)
            for number, (a, b) in enumerate(runs, start=1):
                members = seq[a:b]
                if self.trim:
                    while members and not members[0].protein.domains:
                        members = members[1:]
                    while members and not members[-1].protein.domains:
                        members = members[:-1]
                cluster = Cluster(f"{seq_id}_cluster_{number}", list(members))
                if self._valid(seq, cluster):
                    yield cluster

    def _valid(self, seq: List[Any], cluster: Any) -> bool:
        members = cluster.genes
        if self.criterion == "gecco":
            annotated = sum(1 for g in members if g.protein.domains)
            edge_ids = set()
            if self.edge_distance > 0:
                ids = [g.id for g in seq if g.protein.domains]
                edge_ids = set(ids[: self.edge_distance]) | set(ids[-self.edge_distance:])
            inner = len({g.id for g in members} - edge_ids)
            return annotated >= self.n_cds and inner >= self.n_cds
        domains = {d.name for g in members for d in g.protein.domains}
        ps = [g.average_probability for g in members]
        mean_p = float(numpy.mean(ps)) if ps else float("nan")  # numpy.mean like refine.py:160
        return (mean_p >= self.average_threshold and len(domains & BIO_PFAMS) >= self.n_biopfams
                and len(members) >= self.n_cds)
