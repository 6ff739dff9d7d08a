"""TSV wire format on either side of the CRF path, without polars.

Schemas follow ``/root/reference/gecco/model.py``: ``FeatureTable`` ``:629-642`` (one row per
domain hit), ``GeneTable`` ``:781-789`` (one row per gene), ``ClusterTable`` ``:712-725``;
writing rules follow ``gecco/_base.py:133-152`` (columns holding only their default are
dropped, NaN is written as an empty field) and ``model.py:762-770`` (ClusterTable always
writes every column).  These are the formats of the reference's golden fixtures
(``tests/test_cli/data/BGC0001866.*.tsv``) and of ``gecco predict``'s inputs.
"""
import csv
import math
from typing import Any, Dict, Iterable, List, Optional, Sequence, TextIO, Union

import numpy as np

from . import model as _model

try:  # bulk tables (metagenomes: millions of rows) go through pandas' C parser / writer
    import pandas as _pd
except ImportError:  # pragma: no cover - pandas is optional
    _pd = None

_FEATURE_COLUMNS = [
    ("sequence_id", str, None), ("protein_id", str, None), ("start", int, None), ("end", int, None),
    ("strand", str, None), ("domain", str, None), ("hmm", str, None), ("i_evalue", float, None),
    ("pvalue", float, None), ("domain_start", int, None), ("domain_end", int, None),
    ("cluster_probability", float, math.nan),
]
_GENE_COLUMNS = [
    ("sequence_id", str, None), ("protein_id", str, None), ("start", int, None), ("end", int, None),
    ("strand", str, None), ("average_p", float, math.nan), ("max_p", float, math.nan),
]


def _fmt(v: Any) -> str:
    if v is None:
        return ""
    if isinstance(v, (float, np.floating)):
        return "" if math.isnan(v) else repr(float(v))
    return str(v)


def _parse(text: str, typ: type) -> Any:
    if typ is float:
        return math.nan if text == "" else float(text)
    if typ is int:
        return int(text)
    return text


class _Table:
    COLUMNS: List[tuple] = []

    def __init__(self, columns: Optional[Dict[str, List[Any]]] = None):
        self.columns: Dict[str, List[Any]] = columns or {name: [] for name, _, _ in self.COLUMNS}
        n = len(self)
        for name, _, default in self.COLUMNS:
            if name not in self.columns:
                self.columns[name] = [default] * n

    def __len__(self) -> int:
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def __getattr__(self, name: str) -> List[Any]:
        try:
            return self.__dict__["columns"][name]
        except KeyError as err:
            raise AttributeError(name) from err

    @classmethod
    def load(cls, fh: Union[str, TextIO]):
        """Columns come back as numpy arrays (object arrays for text) when pandas is available,
        as lists otherwise; floats are parsed correctly rounded either way."""
        types = {name: typ for name, typ, _ in cls.COLUMNS}
        if _pd is not None:
            dt = {name: (np.float64 if typ is float else np.int64 if typ is int else str) for name, typ in types.items()}
            df = _pd.read_csv(fh, sep="\t", dtype=dt, keep_default_na=False,
                              na_values={name: [""] for name, typ in types.items() if typ is float},
                              float_precision="round_trip", quoting=csv.QUOTE_NONE)
            return cls({h: df[h].to_numpy(dtype=object) if df[h].dtype == object else df[h].to_numpy() for h in df.columns})
        own = isinstance(fh, str)
        f = open(fh, newline="") if own else fh
        try:
            reader = csv.reader(f, delimiter="\t")
            header = next(reader)
            cols: Dict[str, List[Any]] = {h: [] for h in header}
            for row in reader:
                for h, cell in zip(header, row):
                    cols[h].append(_parse(cell, types.get(h, str)))
        finally:
            if own:
                f.close()
        return cls(cols)

    def _dump_columns(self) -> List[str]:
        keep = []
        for name, _, default in self.COLUMNS:
            col = self.columns[name]
            if default is not None and len(col):
                if isinstance(default, float) and math.isnan(default):
                    arr = np.asarray(col, dtype=np.float64) if not isinstance(col, np.ndarray) else col
                    if arr.dtype.kind == "f" and np.isnan(arr).all():
                        continue
                elif all(v == default for v in col):
                    continue
            keep.append(name)
        return keep

    def dump(self, fh: Union[str, TextIO]) -> None:
        names = self._dump_columns()
        if _pd is not None and len(self) > 64:
            # floats are written with repr() digits (shortest round trip), NaN as an empty field
            _pd.DataFrame({n: self.columns[n] for n in names}, columns=names).to_csv(
                fh, sep="\t", index=False, na_rep="", quoting=csv.QUOTE_NONE, lineterminator="\n")
            return
        own = isinstance(fh, str)
        f = open(fh, "w", newline="") if own else fh
        try:
            f.write("\t".join(names) + "\n")
            for i in range(len(self)):
                f.write("\t".join(_fmt(self.columns[n][i]) for n in names) + "\n")
        finally:
            if own:
                f.close()


class FeatureTable(_Table):
    COLUMNS = _FEATURE_COLUMNS

    @classmethod
    def from_genes(cls, genes: Iterable[Any]) -> "FeatureTable":
        cols: Dict[str, List[Any]] = {name: [] for name, _, _ in cls.COLUMNS}
        for gene in genes:
            for d in gene.protein.domains:
                cols["sequence_id"].append(gene.source.id)
                cols["protein_id"].append(gene.protein.id)
                cols["start"].append(gene.start)
                cols["end"].append(gene.end)
                cols["strand"].append(gene.strand.sign)
                cols["domain"].append(d.name)
                cols["hmm"].append(d.hmm)
                cols["i_evalue"].append(d.i_evalue)
                cols["pvalue"].append(d.pvalue)
                cols["domain_start"].append(d.start)
                cols["domain_end"].append(d.end)
                cols["cluster_probability"].append(math.nan if d.probability is None else d.probability)
        return cls(cols)

    def to_genes(self) -> List[Any]:
        """One Gene per distinct protein_id (sorted by id, like model.py:679), with its domains."""
        rows: Dict[str, List[int]] = {}
        for i, pid in enumerate(self.protein_id):
            rows.setdefault(pid, []).append(i)
        genes = []
        for pid in sorted(rows):
            idx = rows[pid]
            i0 = idx[0]
            strand = _model.Strand.Coding if self.strand[i0] == "+" else _model.Strand.Reverse
            protein = _model.Protein(pid, None)
            gene = _model.Gene(_model.Source(self.sequence_id[i0]), self.start[i0], self.end[i0], strand, protein)
            for i in idx:
                p = self.cluster_probability[i]
                protein.domains.append(_model.Domain(
                    self.domain[i], self.domain_start[i], self.domain_end[i], self.hmm[i], self.i_evalue[i],
                    self.pvalue[i], None if (isinstance(p, float) and math.isnan(p)) else p))
            genes.append(gene)
        return genes


class GeneTable(_Table):
    COLUMNS = _GENE_COLUMNS

    @classmethod
    def from_genes(cls, genes: Iterable[Any]) -> "GeneTable":
        cols: Dict[str, List[Any]] = {name: [] for name, _, _ in cls.COLUMNS}
        for gene in genes:
            cols["sequence_id"].append(gene.source.id)
            cols["protein_id"].append(gene.protein.id)
            cols["start"].append(gene.start)
            cols["end"].append(gene.end)
            cols["strand"].append(gene.strand.sign)
            ap, mp = gene.average_probability, gene.maximum_probability
            cols["average_p"].append(math.nan if ap is None else ap)
            cols["max_p"].append(math.nan if mp is None else mp)
        return cls(cols)

    def to_genes(self) -> List[Any]:
        genes = []
        has_p = "average_p" in self.columns
        for i, pid in enumerate(self.protein_id):
            strand = _model.Strand.Coding if self.strand[i] == "+" else _model.Strand.Reverse
            p = self.average_p[i] if has_p else None
            if isinstance(p, float) and math.isnan(p):
                p = None
            genes.append(_model.Gene(_model.Source(self.sequence_id[i]), self.start[i], self.end[i], strand,
                                     _model.Protein(pid, None), _probability=p))
        return genes


class ClusterTable(_Table):
    COLUMNS = [
        ("sequence_id", str, None), ("cluster_id", str, None), ("start", int, None), ("end", int, None),
        ("average_p", float, math.nan), ("max_p", float, math.nan), ("type", str, "Unknown"),
        ("proteins", str, ""), ("domains", str, ""),
    ]

    @classmethod
    def from_clusters(cls, clusters: Iterable[Any]) -> "ClusterTable":
        cols: Dict[str, List[Any]] = {name: [] for name, _, _ in cls.COLUMNS}
        for c in clusters:
            cols["sequence_id"].append(c.source.id)
            cols["cluster_id"].append(c.id)
            cols["start"].append(c.start)
            cols["end"].append(c.end)
            ap, mp = c.average_probability, c.maximum_probability
            cols["average_p"].append(math.nan if ap is None else ap)
            cols["max_p"].append(math.nan if mp is None else mp)
            cols["type"].append("Unknown" if getattr(c, "type", None) is None else str(c.type))
            cols["proteins"].append(";".join(sorted(g.protein.id for g in c.genes)))
            cols["domains"].append(";".join(sorted(d.name for g in c.genes for d in g.protein.domains)))
        return cls(cols)

    def _dump_columns(self) -> List[str]:  # model.py:762-770: every column is always written
        return [name for name, _, _ in self.COLUMNS]
