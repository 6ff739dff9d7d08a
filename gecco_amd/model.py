"""Thin value types mirroring the slice of ``gecco.model`` the CRF hot path touches
(``/root/reference/gecco/model.py:110-196`` Domain, ``:199-237`` Protein, ``:240-387`` Gene,
``:390-454`` Cluster).

They exist so that the path can be exercised and tested where GECCO itself (and Biopython)
is not installed.  ``gecco_amd.crf.ClusterCRF`` never constructs these classes itself: it
only calls the *methods* (`with_probability`, `with_protein`, `with_domains`,
`with_cluster_weight`) of whatever objects it is handed, so genuine ``gecco.model`` objects
flow through unchanged when GECCO is present.
"""
import enum
import statistics
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Optional


class Strand(enum.IntEnum):
    Coding = 1
    Reverse = -1

    @property
    def sign(self) -> str:
        return "+" if self is Strand.Coding else "-"


@dataclass(frozen=True)
class Source:
    """Stand-in for the Biopython ``SeqRecord`` a gene points to (only ``.id`` is used)."""

    id: str
    seq: Any = None


@dataclass(frozen=True)
class Domain:
    name: str
    start: int
    end: int
    hmm: str
    i_evalue: float
    pvalue: float
    probability: Optional[float] = None
    cluster_weight: Optional[float] = None
    go_terms: List[Any] = field(default_factory=list)
    go_functions: List[Any] = field(default_factory=list)
    qualifiers: Dict[str, List[str]] = field(default_factory=dict)

    def with_probability(self, probability: Optional[float]) -> "Domain":
        return Domain(self.name, self.start, self.end, self.hmm, self.i_evalue, self.pvalue, probability,
                      self.cluster_weight, self.go_terms, self.go_functions, self.qualifiers.copy())

    def with_cluster_weight(self, cluster_weight: Optional[float]) -> "Domain":
        return Domain(self.name, self.start, self.end, self.hmm, self.i_evalue, self.pvalue, self.probability,
                      cluster_weight, self.go_terms, self.go_functions, self.qualifiers.copy())


@dataclass(frozen=True)
class Protein:
    id: str
    seq: Any
    domains: List[Domain] = field(default_factory=list)

    def with_domains(self, domains: Iterable[Domain]) -> "Protein":
        return Protein(self.id, self.seq, list(domains))


@dataclass(frozen=True)
class Gene:
    source: Any
    start: int
    end: int
    strand: Strand
    protein: Protein
    qualifiers: Dict[str, List[str]] = field(default_factory=dict)
    _probability: Optional[float] = None

    @property
    def id(self) -> str:
        return self.protein.id

    @property
    def average_probability(self) -> Optional[float]:
        if self._probability is not None:
            return self._probability
        p = [d.probability for d in self.protein.domains if d.probability is not None]
        return statistics.mean(p) if p else None

    @property
    def maximum_probability(self) -> Optional[float]:
        if self._probability is not None:
            return self._probability
        p = [d.probability for d in self.protein.domains if d.probability is not None]
        return max(p) if p else None

    def with_protein(self, protein: Protein) -> "Gene":
        return Gene(self.source, self.start, self.end, self.strand, protein, self.qualifiers.copy(),
                    _probability=self._probability)

    def with_probability(self, probability: float) -> "Gene":
        return Gene(
            self.source, self.start, self.end, self.strand,
            self.protein.with_domains([d.with_probability(probability) for d in self.protein.domains]),
            self.qualifiers.copy(), _probability=probability,
        )


class Cluster:
    """A run of contiguous genes (``gecco/model.py:390-454``)."""

    def __init__(self, id: str, genes: Optional[List[Gene]] = None, type: Any = None,
                 type_probabilities: Optional[Dict[str, float]] = None):
        self.id = id
        self.genes = genes or list()
        self.type = type
        self.type_probabilities = type_probabilities or dict()

    @property
    def source(self) -> Any:
        return self.genes[0].source

    @property
    def start(self) -> int:
        return min(g.start for g in self.genes)

    @property
    def end(self) -> int:
        return max(g.end for g in self.genes)

    @property
    def average_probability(self) -> Optional[float]:
        # statistics.mean is exactly rounded; numpy's pairwise mean is 1 ulp off on the
        # reference's own fixture
        p = [g.average_probability for g in self.genes if g.average_probability is not None]
        return statistics.mean(p) if p else None

    @property
    def maximum_probability(self) -> Optional[float]:
        p = [g.maximum_probability for g in self.genes if g.maximum_probability is not None]
        return max(p) if p else None
