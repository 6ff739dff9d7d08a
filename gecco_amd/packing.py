"""Host-side packing of GECCO's object model into the CSR batch the engine consumes.

Row X of SURVEY.md §8a: ``extract_features_protein`` / ``extract_features_domain``
(``/root/reference/gecco/crf/features.py:13-48``) produce one ``{domain.name: True}`` dict per
gene (protein mode; duplicate names collapse, order = first occurrence) or per domain (domain
mode; genes without domains give one empty item).  [EXT] CRFsuite then looks every key up in
the model's attribute dictionary and silently drops unknown ones.  Here both steps happen at
once: names -> int32 attribute ids in CSR form.
"""
from dataclasses import dataclass
from typing import Any, Dict, List, Sequence

import numpy as np


@dataclass
class PackedBatch:
    item_ptr: np.ndarray  # [n_contigs+1] item (gene or domain) offsets per contig, int64
    attr_ptr: np.ndarray  # [n_items+1]   attribute offsets per item, int64
    attr_id: np.ndarray   # [nnz]         int32 attribute ids known to the model


def pack_contigs(contigs: Sequence[Sequence[Any]], attr_index: Dict[str, int], feature_type: str = "protein") -> PackedBatch:
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


def pack_columns(sequence_id: Sequence[str], protein_id: Sequence[str], start: Sequence[int], domain: Sequence[str],
                 domain_start: Sequence[int], attr_index: Dict[str, int], gene_sequence_id: Sequence[str] = None,
                 gene_protein_id: Sequence[str] = None, gene_start: Sequence[int] = None):
    """Columnar packer: FeatureTable columns (one row per domain hit,
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
