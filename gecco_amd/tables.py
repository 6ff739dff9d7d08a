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


try:
    import pyarrow as _pa
except ImportError:  # pragma: no cover - pyarrow is optional
    _pa = None


class StringColumn:
    """A text column in Arrow layout: one utf-8 byte buffer + int64 offsets (``offsets[i]:offsets[i+1]`` is row
    i).  This is what the native packer (``gecco_crf_pack_columns``) consumes without touching a Python
    object per row; everything else sees a sequence of ``str`` (``__getitem__``, iteration, ``__array__``)."""

    __slots__ = ("data", "offsets", "_objects")

    def __init__(self, data: np.ndarray, offsets: np.ndarray, objects: Optional[np.ndarray] = None):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self._objects = objects

    @classmethod
    def from_sequence(cls, seq: Sequence[Any]) -> "StringColumn":
        if isinstance(seq, StringColumn):
            return seq
        objects = seq if isinstance(seq, np.ndarray) and seq.dtype == object else None
        if _pa is not None:
            arr = seq if isinstance(seq, (_pa.Array, _pa.ChunkedArray)) else _pa.array(seq, type=_pa.large_string())
            return cls.from_arrow(arr, objects)
        enc = [str(x).encode("utf-8") for x in seq]
        off = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in enc], out=off[1:])
        return cls(np.frombuffer(b"".join(enc), dtype=np.uint8), off, objects)

    @classmethod
    def from_arrow(cls, arr: Any, objects: Optional[np.ndarray] = None) -> "StringColumn":
        if isinstance(arr, _pa.ChunkedArray):
            arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
        if arr.type != _pa.large_string():
            arr = arr.cast(_pa.large_string())
        if arr.null_count:
            arr = arr.fill_null("")
        _, off_buf, data_buf = arr.buffers()
        off = np.frombuffer(off_buf, dtype=np.int64)[arr.offset:arr.offset + len(arr) + 1]
        data = np.frombuffer(data_buf, dtype=np.uint8) if data_buf is not None else np.zeros(0, dtype=np.uint8)
        if len(off) and off[0] != 0:  # a slice: rebase
            data, off = data[off[0]:off[-1]], off - off[0]
        if len(off) == 0:
            off = np.zeros(1, dtype=np.int64)
        return cls(data, off, objects)

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def to_objects(self) -> np.ndarray:
        if self._objects is None:
            raw = self.data.tobytes()
            off = self.offsets.tolist()
            self._objects = np.array([raw[a:b].decode("utf-8") for a, b in zip(off[:-1], off[1:])] or [], dtype=object)
        return self._objects

    def __array__(self, dtype=None, copy=None):
        obj = self.to_objects()
        return obj if dtype in (None, object) else obj.astype(dtype)

    def __iter__(self):
        return iter(self.to_objects())

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            if self._objects is not None:
                return self._objects[i]
            a, b = int(self.offsets[i]), int(self.offsets[i + 1])
            return self.data[a:b].tobytes().decode("utf-8")
        return self.take(np.arange(len(self))[i])

    def take(self, idx: np.ndarray) -> "StringColumn":
        idx = np.asarray(idx, dtype=np.int64)
        lens = self.offsets[idx + 1] - self.offsets[idx]
        off = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        src = np.repeat(self.offsets[idx] - off[:-1], lens) + np.arange(int(off[-1]), dtype=np.int64)
        return StringColumn(self.data[src], off, None if self._objects is None else self._objects[idx])

    def tolist(self) -> List[str]:
        return self.to_objects().tolist()


def as_string_column(col: Any) -> StringColumn:
    return col if isinstance(col, StringColumn) else StringColumn.from_sequence(col)


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

    def string_column(self, name: str) -> "StringColumn":
        """The column in Arrow layout (converted once and kept: bulk tables are handed to the native packer)."""
        col = self.columns[name]
        if not isinstance(col, StringColumn):
            col = self.columns[name] = StringColumn.from_sequence(col)
        return col

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
        if _pa is not None:
            # Arrow's multi-threaded reader: text columns stay in Arrow layout (no Python object per cell),
            # floats are parsed correctly rounded, an empty numeric cell is NaN, an empty text cell is ""
            import pyarrow.csv as _pcsv

            tmap = {name: (_pa.float64() if typ is float else _pa.int64() if typ is int else _pa.large_string())
                    for name, typ in types.items()}
            tab = _pcsv.read_csv(
                fh, parse_options=_pcsv.ParseOptions(delimiter="\t", quote_char=False),
                convert_options=_pcsv.ConvertOptions(column_types=tmap, null_values=[""], strings_can_be_null=False))
            cols: Dict[str, Any] = {}
            for h in tab.column_names:
                c = tab.column(h)
                if _pa.types.is_large_string(c.type) or _pa.types.is_string(c.type):
                    cols[h] = StringColumn.from_arrow(c)
                elif _pa.types.is_floating(c.type):
                    cols[h] = c.to_numpy(zero_copy_only=False).astype(np.float64, copy=False)
                elif _pa.types.is_integer(c.type):
                    if c.null_count:  # the row parser's int("") (a null would come out of astype as INT64_MIN)
                        raise ValueError(f"invalid literal for int() with base 10: '' (column {h!r})")
                    cols[h] = c.to_numpy(zero_copy_only=False).astype(np.int64, copy=False)
                else:
                    cols[h] = np.array(c.to_pylist(), dtype=object)
            return cls(cols)
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
        if len(self) > 64:
            # bulk tables: formatted natively (gecco_crf_tsv_format: several host threads, floats with repr()
            # digits, NaN as an empty field) -- the same bytes the row-by-row writer below produces
            try:
                from . import _native

                types = {name: typ for name, typ, _ in self.COLUMNS}
                cols = []
                for n in names:
                    col = self.columns[n]
                    if types.get(n, str) is str or isinstance(col, StringColumn):
                        sc = as_string_column(col)
                        cols.append((sc.data, sc.offsets))
                    else:
                        cols.append(np.asarray(col, dtype=np.float64 if types[n] is float else np.int64))
                text = _native.tsv_format("\t".join(names) + "\n", cols)
            except (ImportError, OSError, TypeError, ValueError):
                text = None  # (a None in a numeric column, ...: the row writer below copes)
            if text is not None:
                if isinstance(fh, str):
                    with open(fh, "wb") as f:
                        f.write(text)
                else:
                    fh.write(text.decode("utf-8"))
                return
        if _pd is not None and len(self) > 64:
            # floats are written with repr() digits (shortest round trip), NaN as an empty field
            _pd.DataFrame({n: (self.columns[n].to_objects() if isinstance(self.columns[n], StringColumn) else self.columns[n])
                           for n in names}, columns=names).to_csv(
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
