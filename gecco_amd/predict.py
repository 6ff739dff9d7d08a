"""Columnar ``gecco predict``: TSV tables in, TSV tables out, no ``Gene`` objects in between.

SURVEY.md §8f rank 1.  Mirrors the data flow of ``/root/reference/gecco/cli/commands/predict.py``
(``:63-88`` load genes + features tables, ``:91-93`` ``predict_probabilities``, ``:102-110``
``extract_clusters``, then the genes / features / clusters tables are written) but keeps
everything in columns: FeatureTable/GeneTable columns -> CSR (``packing.pack_columns``) ->
windowed marginals on the device -> cluster segmentation on the device
(``gecco_crf_segment``) -> output columns.  Type classification (the ``type`` /
``*_probability`` columns of clusters.tsv) is out of scope and written as ``Unknown``.

    python -m gecco_amd.predict --genes X.genes.tsv --features X.features.tsv --model DIR -o OUT
"""
import argparse
import math
import os
import statistics
import sys
import warnings
from typing import List, Optional

import numpy as np

from . import _native, composition, packing, tables
from .crf import ClusterCRF


def predict_tables(genes_t: tables.GeneTable, feats_t: tables.FeatureTable, crf: ClusterCRF, *, pad: bool = True,
                   threshold: float = 0.8, n_cds: int = 3, edge_distance: int = 0, trim: bool = True,
                   device: Optional[int] = None, composition_domains: Optional[List[str]] = None):
    """Returns (GeneTable, FeatureTable, ClusterTable) with probabilities / clusters filled in; with
    `composition_domains` (the type classifier's domain list) also the (n_clusters, n_domains)
    weighted domain composition matrix ``TypeClassifier.predict_types`` feeds its forest with."""
    if crf.feature_type != "protein":
        raise ValueError("the columnar path supports protein-level features (the shipped model's mode)")
    dev = crf.devices[0] if device is None else device
    idx = crf.model._attr_index
    contig_ids, order, cptr, gptr, attr, annotated = packing.pack_columns(
        feats_t.sequence_id, feats_t.protein_id, feats_t.start, feats_t.domain, feats_t.domain_start, idx,
        genes_t.sequence_id, genes_t.protein_id, genes_t.start)
    W = crf.window_size
    for c, cid in enumerate(contig_ids):  # the reference's warnings (crf/__init__.py:216-233)
        n = int(cptr[c + 1] - cptr[c])
        if n < W:
            if pad:
                unit = "protein" if W - n == 1 else "proteins"
                warnings.warn(f"Contig {cid!r} does not contain enough proteins ({n}) for sliding window of size {W}, "
                              f"padding with {W - n} {unit}")
            else:
                warnings.warn(f"Contig {cid!r} does not contain enough proteins ({n}) for sliding window of size {W}")
    p = crf.predict_probabilities_csr(cptr, gptr, attr, pad=pad, device=dev)
    seg = _native.segment(p, annotated, cptr, threshold, n_cds, edge_distance, trim, device=dev)

    # ---- genes table, in the order of ClusterCRF.predict_probabilities (contig id, start)
    row_of = {pid: i for i, pid in enumerate(genes_t.protein_id)}
    have_row = all(pid in row_of for pid in order)
    gcols = {name: [] for name, _, _ in tables.GeneTable.COLUMNS}
    p_of = {}
    for k, pid in enumerate(order):
        pv = float(p[k])
        p_of[pid] = pv
        if have_row:
            i = row_of[pid]
            gcols["sequence_id"].append(genes_t.sequence_id[i])
            gcols["protein_id"].append(pid)
            gcols["start"].append(genes_t.start[i])
            gcols["end"].append(genes_t.end[i])
            gcols["strand"].append(genes_t.strand[i])
        gcols["average_p"].append(pv)
        gcols["max_p"].append(pv)
    genes_out = tables.GeneTable(gcols) if have_row else None

    # ---- features table: every domain row carries its gene's probability (features.py:92-96)
    fcols = {name: list(col) for name, col in feats_t.columns.items()}
    fcols["cluster_probability"] = [p_of.get(pid, math.nan) for pid in feats_t.protein_id]
    feats_out = tables.FeatureTable(fcols)

    # ---- clusters table (gecco/model.py:731-760)
    doms_of = {}
    for pid, dom in zip(feats_t.protein_id, feats_t.domain):
        doms_of.setdefault(pid, []).append(dom)
    ccols = {name: [] for name, _, _ in tables.ClusterTable.COLUMNS}
    for c, number, a, b in seg.tolist():
        members = order[a:b]
        rows = [row_of[pid] for pid in members]
        ps = [p_of[pid] for pid in members if not math.isnan(p_of[pid])]
        ccols["sequence_id"].append(contig_ids[c])
        ccols["cluster_id"].append(f"{contig_ids[c]}_cluster_{number}")
        ccols["start"].append(min(genes_t.start[i] for i in rows))
        ccols["end"].append(max(genes_t.end[i] for i in rows))
        ccols["average_p"].append(statistics.mean(ps) if ps else math.nan)  # exactly rounded, model.py:442-447
        ccols["max_p"].append(max(ps) if ps else math.nan)
        ccols["type"].append("Unknown")
        ccols["proteins"].append(";".join(sorted(members)))
        ccols["domains"].append(";".join(sorted(d for pid in members for d in doms_of.get(pid, ()))))
    clusters_out = tables.ClusterTable(ccols)
    if composition_domains is None:
        return genes_out, feats_out, clusters_out
    # input matrix of the type classifier (types/__init__.py:118), one row per called cluster
    comps = composition.table_compositions(seg, order, feats_t.protein_id, feats_t.domain, feats_t.pvalue,
                                           feats_t.domain_start, composition_domains, device=dev)
    return genes_out, feats_out, clusters_out, comps


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m gecco_amd.predict", description=__doc__.split("\n")[0])
    ap.add_argument("--genes", required=True, help="gene table (gecco annotate / gecco run *.genes.tsv)")
    ap.add_argument("--features", required=True, help="domain annotation table (*.features.tsv)")
    ap.add_argument("--model", default=None, help="model directory (model.pkl + model.pkl.md5); default: GECCO's embedded model")
    ap.add_argument("-o", "--output-dir", default=".")
    ap.add_argument("--no-pad", action="store_true")
    ap.add_argument("-m", "--threshold", type=float, default=0.8)
    ap.add_argument("--cds", type=int, default=3)
    ap.add_argument("-E", "--edge-distance", type=int, default=0)
    ap.add_argument("--no-trim", action="store_true")
    ap.add_argument("--composition-domains", default=None,
                    help="file with one domain accession per line (the type classifier's domains.tsv): also write "
                         "<base>.compositions.npy, the classifier's input matrix")
    args = ap.parse_args(argv)
    crf = ClusterCRF.trained(args.model)
    genes_t = tables.GeneTable.load(args.genes)
    feats_t = tables.FeatureTable.load(args.features)
    comp_domains = None
    if args.composition_domains:
        with open(args.composition_domains) as fh:
            comp_domains = [line.strip() for line in fh if line.strip()]
    res = predict_tables(
        genes_t, feats_t, crf, pad=not args.no_pad, threshold=args.threshold, n_cds=args.cds,
        edge_distance=args.edge_distance, trim=not args.no_trim, composition_domains=comp_domains)
    genes_out, feats_out, clusters = res[:3]
    os.makedirs(args.output_dir, exist_ok=True)
    base = os.path.splitext(os.path.basename(args.genes))[0]
    base = base[:-len(".genes")] if base.endswith(".genes") else base
    if genes_out is not None:
        genes_out.dump(os.path.join(args.output_dir, f"{base}.genes.tsv"))
    feats_out.dump(os.path.join(args.output_dir, f"{base}.features.tsv"))
    clusters.dump(os.path.join(args.output_dir, f"{base}.clusters.tsv"))
    if comp_domains is not None:
        np.save(os.path.join(args.output_dir, f"{base}.compositions.npy"), res[3])
    print(f"{len(genes_t)} genes, {len(clusters)} clusters -> {args.output_dir}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
