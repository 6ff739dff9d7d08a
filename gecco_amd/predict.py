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
    pk = packing.pack_columns(
        feats_t.sequence_id, feats_t.protein_id, feats_t.start, feats_t.domain, feats_t.domain_start, idx,
        genes_t.sequence_id, genes_t.protein_id, genes_t.start)
    contig_ids, order, cptr, gptr, attr, annotated = pk
    W = crf.window_size
    lengths = np.diff(cptr)
    for c in np.flatnonzero(lengths < W):  # the reference's warnings (crf/__init__.py:216-233)
        cid, n = contig_ids[c], int(lengths[c])
        if pad:
            unit = "protein" if W - n == 1 else "proteins"
            warnings.warn(f"Contig {cid!r} does not contain enough proteins ({n}) for sliding window of size {W}, "
                          f"padding with {W - n} {unit}")
        else:
            warnings.warn(f"Contig {cid!r} does not contain enough proteins ({n}) for sliding window of size {W}")
    # marginals and cluster rows in one pass of the batch driver: the refiner runs on the device right behind
    # the marginals (one grouper per contig, like the CLI: cli/commands/_common.py:621-623); the probabilities
    # come back once, for the output tables
    label = crf.model.native.label_id("1")
    if label < 0:
        raise ValueError("the model has no label '1'")
    session = crf._session() if device is None else _native.Session(crf.model.native, [dev])
    seg, _, _, p = session.clusters(cptr, gptr, attr, annotated, W, crf.window_step, label, pad, threshold, n_cds,
                                    edge_distance, trim, want_p=True, want_seg_p=False)

    # ---- genes table, in the order of ClusterCRF.predict_probabilities (contig id, start); every
    # column is gathered at once (no per-gene Python work)
    rows = None
    try:
        import pandas as pd

        index = pd.Index(np.asarray(genes_t.protein_id, dtype=object))
        if index.is_unique:
            rows = index.get_indexer(order).astype(np.int64)
    except ImportError:  # pragma: no cover
        pass
    if rows is None:  # repeated ids: the last row of an id stands for it, like a dict
        row_of = {pid: i for i, pid in enumerate(genes_t.protein_id)}
        rows = np.fromiter((row_of.get(pid, -1) for pid in order), dtype=np.int64, count=len(order))
    have_row = bool((rows >= 0).all())
    genes_out = None
    if have_row:
        gcols = {name: np.asarray(genes_t.columns[name])[rows] for name in ("sequence_id", "protein_id", "start", "end", "strand")}
        gcols["average_p"] = p
        gcols["max_p"] = p
        genes_out = tables.GeneTable(gcols)

    # ---- features table: every domain row carries its gene's probability (features.py:92-96)
    fcols = dict(feats_t.columns)
    fcols["cluster_probability"] = p[pk.row_gene] if len(pk.row_gene) else np.zeros(0)
    feats_out = tables.FeatureTable(fcols)

    # ---- clusters table (gecco/model.py:731-760): a handful of rows
    g_start = np.asarray(genes_t.start)
    g_end = np.asarray(genes_t.end)
    f_domain = np.asarray(feats_t.domain, dtype=object)
    ccols = {name: [] for name, _, _ in tables.ClusterTable.COLUMNS}
    for c, number, a, b in seg.tolist():
        members = [str(x) for x in order[a:b]]
        r = rows[a:b]
        ps = [float(v) for v in p[a:b] if not math.isnan(v)]
        ccols["sequence_id"].append(contig_ids[c])
        ccols["cluster_id"].append(f"{contig_ids[c]}_cluster_{number}")
        ccols["start"].append(int(g_start[r].min()))
        ccols["end"].append(int(g_end[r].max()))
        ccols["average_p"].append(statistics.mean(ps) if ps else math.nan)  # exactly rounded, model.py:442-447
        ccols["max_p"].append(max(ps) if ps else math.nan)
        ccols["type"].append("Unknown")
        ccols["proteins"].append(";".join(sorted(members)))
        drows = pk.row_order[pk.row_ptr[a]:pk.row_ptr[b]]
        ccols["domains"].append(";".join(sorted(str(d) for d in f_domain[drows])))
    clusters_out = tables.ClusterTable(ccols)
    if composition_domains is None:
        return genes_out, feats_out, clusters_out
    # input matrix of the type classifier (types/__init__.py:118), one row per called cluster
    comps = composition.packed_compositions(seg, pk, feats_t.domain, feats_t.pvalue, composition_domains, device=dev)
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
