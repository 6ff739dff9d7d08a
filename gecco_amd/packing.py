"""Host-side packing of GECCO's object model into the CSR batch the engine consumes.

Row X of SURVEY.md §8a: ``extract_features_protein`` / ``extract_features_domain``
(``/root/reference/gecco/crf/features.py:13-48``) produce one ``{domain.name: True}`` dict per
gene (protein mode; duplicate names collapse, order = first occurrence) or per domain (domain
mode; genes without domains give one empty item).  [EXT] CRFsuite then looks every key up in
the model's attribute dictionary and silently drops unknown ones.  Here both steps happen at
once: names -> int32 attribute ids in CSR form.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np


@dataclass
class PackedBatch:
    item_ptr: np.ndarray  # [n_contigs+1] item (gene or domain) offsets per contig, int64
    attr_ptr: np.ndarray  # [n_items+1]   attribute offsets per item, int64
    attr_id: np.ndarray   # [nnz]         int32 attribute ids known to the model


def pack_contigs(contigs: Sequence[Sequence[Any]], attr_index: Dict[str, int], feature_type: str = "protein") -> PackedBatch:
    if feature_type == "protein":
        from ._objpath_loader import module

        native = module()  # csrc/objpath.c: the loop below against the CPython C API
        if native is not None:
            ip, ap, at = native.pack_protein(contigs, attr_index)
            return PackedBatch(np.frombuffer(ip, dtype=np.int64), np.frombuffer(ap, dtype=np.int64), np.frombuffer(at, dtype=np.int32))
    item_ptr: List[int] = [0]
    attr_ptr: List[int] = [0]
    attr: List[int] = []
    n_items = 0
    get = attr_index.get
    if feature_type == "protein":
        for contig in contigs:
            for gene in contig:
                seen = set()
                for domain in gene.protein.domains:
                    name = domain.name
                    if name in seen:
                        continue  # dict keys: a repeated domain is one feature
                    seen.add(name)
                    idx = get(name)
                    if idx is not None:
                        attr.append(idx)
                attr_ptr.append(len(attr))
            n_items += len(contig)
            item_ptr.append(n_items)
    elif feature_type == "domain":
        for contig in contigs:
            for gene in contig:
                domains = gene.protein.domains
                if domains:
                    for domain in domains:
                        idx = get(domain.name)
                        if idx is not None:
                            attr.append(idx)
                        attr_ptr.append(len(attr))
                        n_items += 1
                else:
                    attr_ptr.append(len(attr))
                    n_items += 1
            item_ptr.append(n_items)
    else:
        raise ValueError(f"invalid feature type: {feature_type!r}")
    return PackedBatch(
        np.asarray(item_ptr, dtype=np.int64), np.asarray(attr_ptr, dtype=np.int64), np.asarray(attr, dtype=np.int32)
    )


@dataclass
class PackedColumns:
    """Result of `pack_columns`; unpacks like the 6-tuple (contig_ids, order, contig_ptr, gene_ptr,
    attr_id, annotated).  The extra fields let table writers stay columnar."""
    contig_ids: List[str]
    order: Any              # protein ids in scoring order (contig id, start)
    contig_ptr: np.ndarray  # int32 [n_contigs+1]
    gene_ptr: np.ndarray    # int32 [n_genes+1]
    attr_id: np.ndarray     # int32 [nnz]
    annotated: np.ndarray   # uint8 [n_genes]: the gene has at least one feature row
    row_gene: np.ndarray = None   # int64 [n_rows]: position in `order` of every feature row's gene
    row_order: np.ndarray = None  # int64 [n_rows]: feature rows sorted by (gene position, domain_start), stable
    row_ptr: np.ndarray = None    # int64 [n_genes+1]: offsets of every gene's rows in `row_order`

    def __iter__(self):
        return iter((self.contig_ids, self.order, self.contig_ptr, self.gene_ptr, self.attr_id, self.annotated))


def pack_columns(sequence_id: Sequence[str], protein_id: Sequence[str], start: Sequence[int], domain: Sequence[str],
                 domain_start: Sequence[int], attr_index: Dict[str, int], gene_sequence_id: Sequence[str] = None,
                 gene_protein_id: Sequence[str] = None, gene_start: Sequence[int] = None) -> PackedColumns:
    """Columnar packer: FeatureTable columns (one row per domain hit,
    ``gecco/model.py:629-642``) plus, optionally, GeneTable columns (one row per gene, so that
    genes without any domain are kept) -> CSR batch without materialising Gene objects.

    Order matches ``ClusterCRF.predict_probabilities``: genes by (sequence_id, start) with ties in
    first-appearance order (``sorted`` is stable, crf/__init__.py:204-206), a gene's domains by
    domain_start (:200-201), duplicate names collapsed (features.py:31-35), unknown names dropped
    ([EXT] CRFsuite attribute lookup).  Vectorised (hash factorisation + stable integer sorts):
    a 2 M-gene metagenome packs in about a second; `pack_columns_py` is the row-by-row statement."""
    try:
        import pandas as pd
    except ImportError:  # pragma: no cover
        return PackedColumns(*pack_columns_py(sequence_id, protein_id, start, domain, domain_start, attr_index,
                                              gene_sequence_id, gene_protein_id, gene_start))
    f_sid = np.asarray(sequence_id, dtype=object)
    f_pid = np.asarray(protein_id, dtype=object)
    f_start = np.asarray(start, dtype=np.int64)
    f_dom = np.asarray(domain, dtype=object)
    f_ds = np.asarray(domain_start, dtype=np.int64)
    nf = len(f_pid)
    if gene_protein_id is not None:
        g_pid = np.asarray(gene_protein_id, dtype=object)
        g_sid = np.asarray(gene_sequence_id, dtype=object)
        g_start = np.asarray(gene_start, dtype=np.int64)
    else:
        g_pid, g_sid, g_start = f_pid[:0], f_sid[:0], f_start[:0]
    ng = len(g_pid)
    # genes in first-appearance order: gene-table rows, then proteins only the feature table knows
    codes, uniq = pd.factorize(np.concatenate([g_pid, f_pid]), sort=False)
    n = len(uniq)
    sid_u = np.empty(n, dtype=object)
    start_u = np.zeros(n, dtype=np.int64)
    if nf:  # feature rows: the first row of a protein defines it ...
        sid_u[codes[ng:][::-1]] = f_sid[::-1]
        start_u[codes[ng:][::-1]] = f_start[::-1]
    if ng:  # ... unless the gene table lists it (a repeated id keeps its last row, like a dict)
        sid_u[codes[:ng]] = g_sid
        start_u[codes[:ng]] = g_start
    sid_code, sid_names = pd.factorize(sid_u, sort=True)  # codes follow str ordering
    perm = np.lexsort((start_u, sid_code))                # stable
    rank = np.empty(n, dtype=np.int64)
    rank[perm] = np.arange(n)
    order = uniq[perm]
    sc = sid_code[perm]
    cuts = np.flatnonzero(np.diff(sc)) + 1 if n else np.zeros(0, dtype=np.int64)
    contig_ptr = np.concatenate([[0], cuts, [n]]).astype(np.int32) if n else np.zeros(1, dtype=np.int32)
    contig_ids = [str(x) for x in sid_names[sc[contig_ptr[:-1]]]] if n else []
    # feature rows by (gene position, domain_start), stable
    row_gene = rank[codes[ng:]]
    row_order = np.lexsort((f_ds, row_gene)) if nf else np.zeros(0, dtype=np.int64)
    rows_per_gene = np.bincount(row_gene, minlength=n) if nf else np.zeros(n, dtype=np.int64)
    row_ptr = np.concatenate([[0], np.cumsum(rows_per_gene)]).astype(np.int64)
    dom_code, dom_names = pd.factorize(f_dom, sort=False)
    attr_of_dom = np.fromiter((attr_index.get(d, -1) for d in dom_names), dtype=np.int64, count=len(dom_names))
    g_sorted = row_gene[row_order]
    d_sorted = dom_code[row_order]
    first = ~pd.Series(g_sorted * max(len(dom_names), 1) + d_sorted).duplicated().to_numpy() if nf else np.zeros(0, dtype=bool)
    a_sorted = attr_of_dom[d_sorted] if nf else np.zeros(0, dtype=np.int64)
    keep = first & (a_sorted >= 0)
    attr = a_sorted[keep].astype(np.int32)
    per_gene = np.bincount(g_sorted[keep], minlength=n) if nf else np.zeros(n, dtype=np.int64)
    gene_ptr = np.concatenate([[0], np.cumsum(per_gene)]).astype(np.int32)
    annotated = (rows_per_gene > 0).astype(np.uint8)
    return PackedColumns(contig_ids, order, contig_ptr, gene_ptr, attr, annotated, row_gene, row_order, row_ptr)


def pack_tables(native_model: Any, feats_t: Any, genes_t: Any = None, markers: Optional[Sequence[str]] = None):
    """Native columnar packer (``gecco_crf_pack_columns``, csrc/crf_tables.cpp): the same result as `pack_columns`
    from tables whose text columns are in Arrow layout (``tables.StringColumn``), without a Python object per
    row -- strings are hashed at most once, orders are checked before anything is sorted, and the CSR lands in
    pinned memory ready for the batch driver.  Returns a ``_native.PackedTables``.  With `markers` (domain names,
    at most 256: the antismash criterion's biosynthetic Pfams, refine.py:157-163) it also carries, per gene, which
    of them occur among ALL the gene's domains (`marker_ptr`, `marker_id`)."""
    from . import _native
    from .tables import StringColumn

    g = (None, None, None)
    if genes_t is not None:
        g = (genes_t.string_column("sequence_id"), genes_t.string_column("protein_id"), genes_t.start)
    return _native.PackedTables(
        native_model, feats_t.string_column("sequence_id"), feats_t.string_column("protein_id"), feats_t.start,
        feats_t.string_column("domain"), feats_t.domain_start, *g,
        markers=StringColumn.from_sequence(list(markers)) if markers else None)


def pack_columns_py(sequence_id: Sequence[str], protein_id: Sequence[str], start: Sequence[int], domain: Sequence[str],
                 domain_start: Sequence[int], attr_index: Dict[str, int], gene_sequence_id: Sequence[str] = None,
                 gene_protein_id: Sequence[str] = None, gene_start: Sequence[int] = None):
    """Row-by-row statement of `pack_columns` (the executable specification its tests compare the
    vectorised version with).  Columnar packer: FeatureTable columns (one row per domain hit,
    ``gecco/model.py:629-642``) plus, optionally, GeneTable columns (one row per gene, so that
    genes without any domain are kept) -> (contig ids, gene ids per contig order, contig_ptr,
    gene_ptr, attr_id, annotated) without materialising Gene objects.

    Order matches ``ClusterCRF.predict_probabilities``: genes by (sequence_id, start), a gene's
    domains by domain_start, duplicates collapsed, unknown domains dropped."""
    genes: Dict[str, tuple] = {}
    if gene_protein_id is not None:
        for sid, pid, st in zip(gene_sequence_id, gene_protein_id, gene_start):
            genes[pid] = (sid, int(st))
    hits: Dict[str, List[tuple]] = {}
    for sid, pid, st, dom, ds in zip(sequence_id, protein_id, start, domain, domain_start):
        if pid not in genes:
            genes[pid] = (sid, int(st))
        hits.setdefault(pid, []).append((int(ds), dom))
    order = sorted(genes, key=lambda pid: (genes[pid][0], genes[pid][1]))  # stable, like sorted() in the reference
    contig_ids: List[str] = []
    contig_ptr: List[int] = [0]
    gene_ptr: List[int] = [0]
    attr: List[int] = []
    annotated: List[int] = []
    last = None
    for pid in order:
        sid = genes[pid][0]
        if sid != last:
            if last is not None:
                contig_ptr.append(len(gene_ptr) - 1)
            contig_ids.append(sid)
            last = sid
        rows = sorted(hits.get(pid, ()), key=lambda r: r[0])
        seen = set()
        for _, dom in rows:
            if dom in seen:
                continue
            seen.add(dom)
            idx = attr_index.get(dom)
            if idx is not None:
                attr.append(idx)
        gene_ptr.append(len(attr))
        annotated.append(1 if rows else 0)
    if order:
        contig_ptr.append(len(gene_ptr) - 1)
    return (
        contig_ids, order, np.asarray(contig_ptr, dtype=np.int32), np.asarray(gene_ptr, dtype=np.int32),
        np.asarray(attr, dtype=np.int32), np.asarray(annotated, dtype=np.uint8),
    )
