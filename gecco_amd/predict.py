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
from .refine import BIO_PFAMS


def filter_features(feats_t: tables.FeatureTable, e_filter: Optional[float] = None,
                    p_filter: Optional[float] = None) -> tables.FeatureTable:
    """``filter_domains`` of the reference CLI (cli/commands/_common.py:419-448) on table rows: keep the domain
    hits with ``i_evalue < e_filter`` and ``pvalue < p_filter`` (strict, a cutoff of None keeps everything)."""
    n = len(feats_t)
    keep = np.ones(n, dtype=bool)
    if e_filter is not None:
        keep &= np.asarray(feats_t.i_evalue, dtype=np.float64) < e_filter
    if p_filter is not None:
        keep &= np.asarray(feats_t.pvalue, dtype=np.float64) < p_filter
    if bool(keep.all()):
        return feats_t
    idx = np.flatnonzero(keep)
    cols = {}
    for name, col in feats_t.columns.items():
        cols[name] = col.take(idx) if isinstance(col, tables.StringColumn) else np.asarray(col)[idx]
    return tables.FeatureTable(cols)


def _refiner_order_differs(sid_code: np.ndarray, start: np.ndarray, end: np.ndarray) -> bool:
    """The CRF scores genes in (contig, start) order (crf/__init__.py:199), the refiner walks them in
    (contig, start, end) order (refine.py:190): they differ only if two genes of a contig share a start and
    their ends come in decreasing order."""
    if len(start) < 2:
        return False
    tie = (sid_code[1:] == sid_code[:-1]) & (start[1:] == start[:-1])
    return bool(np.any(tie & (end[1:] < end[:-1])))


def predict_tables(genes_t: tables.GeneTable, feats_t: tables.FeatureTable, crf: ClusterCRF, *, pad: bool = True,
                   threshold: float = 0.8, n_cds: int = 3, edge_distance: int = 0, trim: bool = True,
                   device: Optional[int] = None, composition_domains: Optional[List[str]] = None,
                   criterion: str = "gecco", n_biopfams: int = 5, average_threshold: float = 0.6):
    """Returns (GeneTable, FeatureTable, ClusterTable) with probabilities / clusters filled in; with
    `composition_domains` (the type classifier's domain list) also the (n_clusters, n_domains)
    weighted domain composition matrix ``TypeClassifier.predict_types`` feeds its forest with.

    Columns in, columns out: the native packer turns the tables into a CSR batch (pinned memory), the batch
    driver runs marginals and refiner on the device (one grouper per contig, like the CLI:
    cli/commands/_common.py:621-623), the probabilities come back once for the gene / feature tables, and
    the cluster rows are assembled natively from the members' probabilities.  `criterion` is the CLI's hidden
    ``--postproc`` (cli/commands/_parser.py:294-301 -> refine.py:142-165); "antismash" runs on the device too: the
    packer marks, per gene, which of the biosynthetic Pfams (refine.BIO_PFAMS) occur among its domains."""
    if criterion not in _native.CRITERIA:
        raise ValueError(f"Unknown cluster filtering criterion: {criterion}")
    if crf.feature_type != "protein":
        raise ValueError("the columnar path supports protein-level features (the shipped model's mode)")
    dev = crf.devices[0] if device is None else device
    native = crf.model.native
    markers = sorted(BIO_PFAMS) if criterion == "antismash" else None
    pk = packing.pack_tables(native, feats_t, genes_t, markers)
    refine_kw = dict(criterion=criterion, n_biopfams=n_biopfams, average_threshold=average_threshold)
    # the reference refuses tables that do not describe the same genes (annotate_genes raises KeyError /
    # ValueError, cli/commands/_common.py): a protein the gene table does not list, a repeated gene id
    if genes_t is not None and len(genes_t):
        if pk.n_duplicate_gene_ids:
            raise ValueError(f"{pk.n_duplicate_gene_ids} duplicated protein ids in the gene table")
        if pk.n_unlisted_proteins:
            raise ValueError(f"{pk.n_unlisted_proteins} proteins of the feature table are missing from the gene table")
    n = pk.n_genes
    cptr = pk.contig_ptr
    W = crf.window_size
    rows = pk.gene_row
    have_row = bool(n == 0 or rows.min() >= 0)
    g_sid = genes_t.string_column("sequence_id") if have_row and n else None

    def contig_id(c: int) -> str:
        r = int(rows[cptr[c]])
        return g_sid[r] if r >= 0 else feats_t.string_column("sequence_id")[-1 - r]

    lengths = np.diff(cptr)
    for c in np.flatnonzero(lengths < W):  # the reference's warnings (crf/__init__.py:216-233)
        cid, k = contig_id(int(c)), int(lengths[c])
        if pad:
            unit = "protein" if W - k == 1 else "proteins"
            warnings.warn(f"Contig {cid!r} does not contain enough proteins ({k}) for sliding window of size {W}, "
                          f"padding with {W - k} {unit}")
        else:
            warnings.warn(f"Contig {cid!r} does not contain enough proteins ({k}) for sliding window of size {W}")
    label = native.label_id("1")
    if label < 0:
        raise ValueError("the model has no label '1'")
    if device is None:
        session = crf._session()
    else:  # a session of its own on the named device, in the class's mode (reference bits or fast kernels): same bits either way
        session = _native.Session(native, [dev])
        session.set_reference_bits(crf._reference_bits_now())
    g_end_all = np.asarray(genes_t.end, dtype=np.int64) if genes_t is not None and len(genes_t) else None
    f_end_all = np.asarray(feats_t.end, dtype=np.int64) if len(feats_t) else None
    # refiner order = CRF order unless equal starts come with decreasing ends (rare): then the slow path below.  Both questions are
    # one native pass over the genes on several threads (they were four numpy passes: 1.5 ms of the level's 10 per 0.4 M genes)
    in_order, reorder = have_row and n == 0, None
    if have_row and n:
        g_start_all = np.asarray(genes_t.start, dtype=np.int64)
        if os.environ.get("GECCO_AMD_TABLES_NUMPY_PASSES") == "1":  # A/B switch: the numpy passes of round 5
            in_order = len(genes_t) == n and rows[0] == 0 and bool(np.all(np.diff(rows) == 1))
            differs = _refiner_order_differs(np.repeat(np.arange(pk.n_contigs), lengths), g_start_all if in_order else g_start_all[rows],
                                             g_end_all if in_order else g_end_all[rows])
        else:
            in_order, differs = pk.order_info(g_start_all, g_end_all)
        if differs:
            g_start_o = g_start_all if in_order else g_start_all[rows]
            g_end_o = g_end_all if in_order else g_end_all[rows]
            code = np.repeat(np.arange(pk.n_contigs), lengths)
            assert _refiner_order_differs(code, g_start_o, g_end_o)
            reorder = np.lexsort((g_end_o, g_start_o, code))
    if reorder is None:
        seg, seg_p, seg_off, p = session.clusters(cptr, pk.gene_ptr, pk.attr_id, pk.annotated, W, crf.window_step, label, pad,
                                                  threshold, n_cds, edge_distance, trim, want_p=True, want_seg_p=True,
                                                  marker_ptr=pk.marker_ptr, marker_id=pk.marker_id, **refine_kw)
    else:
        p = session.windowed_marginals(cptr, pk.gene_ptr, pk.attr_id, W, crf.window_step, label, pad)
        mp = mi = None
        if criterion == "antismash":  # the genes' marker lists in refiner order
            cnt = np.diff(pk.marker_ptr)[reorder]
            mp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            take = np.repeat(pk.marker_ptr[:-1][reorder] - mp[:-1], cnt) + np.arange(int(mp[-1]))
            mi = pk.marker_id[take] if len(take) else np.zeros(0, dtype=np.int32)
        seg = _native.segment(p[reorder], pk.annotated[reorder], cptr, threshold, n_cds, edge_distance, trim, device=dev,
                              marker_ptr=mp, marker_id=mi, **refine_kw)
    if n == 0:
        p = np.zeros(0)

    # ---- genes table, in the order of ClusterCRF.predict_probabilities (contig id, start)
    genes_out = None
    if have_row:
        gcols = {}
        for name in ("sequence_id", "protein_id", "start", "end", "strand"):
            col = genes_t.columns[name]
            if in_order:
                gcols[name] = col
            elif isinstance(col, tables.StringColumn):
                gcols[name] = col.take(rows)
            else:
                gcols[name] = np.asarray(col)[rows]
        gcols["average_p"] = p
        gcols["max_p"] = p
        genes_out = tables.GeneTable(gcols)

    # ---- features table: every domain row carries its gene's probability (features.py:92-96)
    fcols = dict(feats_t.columns)
    if os.environ.get("GECCO_AMD_TABLES_NUMPY_PASSES") == "1":
        fcols["cluster_probability"] = p[pk.row_gene] if pk.n_rows else np.zeros(0)
    else:
        fcols["cluster_probability"] = _native.gather_f64(p, pk.row_gene) if pk.n_rows else np.zeros(0)
    feats_out = tables.FeatureTable(fcols)

    # ---- clusters table (gecco/model.py:731-760)
    if reorder is None:
        cr = pk.cluster_rows(seg, seg_p, seg_off, g_end_all, f_end_all)
        k = len(seg)
        ccols = {
            "sequence_id": tables.StringColumn(*cr["sequence_id"]), "cluster_id": tables.StringColumn(*cr["cluster_id"]),
            "start": cr["start"], "end": cr["end"], "average_p": cr["average_p"], "max_p": cr["max_p"],
            "type": np.full(k, "Unknown", dtype=object), "proteins": tables.StringColumn(*cr["proteins"]),
            "domains": tables.StringColumn(*cr["domains"]),
        }
        clusters_out = tables.ClusterTable(ccols)
    else:
        clusters_out = _cluster_table_slow(seg, reorder, p, pk, genes_t, feats_t, rows, contig_id)
    if composition_domains is None:
        return genes_out, feats_out, clusters_out
    if reorder is not None:
        raise NotImplementedError("compositions for tables whose refiner order differs from the scoring order")
    # input matrix of the type classifier (types/__init__.py:118), one row per called cluster
    comps = composition.packed_compositions(seg, pk, feats_t.domain, feats_t.pvalue, composition_domains, device=dev)
    return genes_out, feats_out, clusters_out, comps


def _cluster_table_slow(seg, order, p, pk, genes_t, feats_t, rows, contig_id) -> tables.ClusterTable:
    """Cluster rows gene by gene, for the rare table whose refiner order is a permutation of the scoring order."""
    g_start, g_end = np.asarray(genes_t.start), np.asarray(genes_t.end)
    g_pid = genes_t.string_column("protein_id")
    f_domain = feats_t.string_column("domain")
    ccols = {name: [] for name, _, _ in tables.ClusterTable.COLUMNS}
    for c, number, a, b in seg.tolist():
        members = order[a:b]
        r = rows[members]
        ps = [float(v) for v in p[members] if not math.isnan(v)]
        cid = contig_id(c)
        ccols["sequence_id"].append(cid)
        ccols["cluster_id"].append(f"{cid}_cluster_{number}")
        ccols["start"].append(int(g_start[r].min()))
        ccols["end"].append(int(g_end[r].max()))
        ccols["average_p"].append(statistics.mean(ps) if ps else math.nan)  # exactly rounded, model.py:442-447
        ccols["max_p"].append(max(ps) if ps else math.nan)
        ccols["type"].append("Unknown")
        ccols["proteins"].append(";".join(sorted(g_pid[int(i)] for i in r)))
        drows = np.concatenate([pk.row_order[pk.row_ptr[g]:pk.row_ptr[g + 1]] for g in members]) if len(members) else []
        ccols["domains"].append(";".join(sorted(f_domain[int(d)] for d in drows)))
    return tables.ClusterTable(ccols)


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
    ap.add_argument("--postproc", choices=["gecco", "antismash"], default="gecco", help=argparse.SUPPRESS)
    ap.add_argument("--no-trim", action="store_true")
    ap.add_argument("-e", "--e-filter", type=float, default=None,
                    help="e-value cutoff for protein domains to be included (gecco predict -e)")
    ap.add_argument("-p", "--p-filter", type=float, default=1e-9,
                    help="p-value cutoff for protein domains to be included (gecco predict -p, default 1e-9)")
    ap.add_argument("--reference-bits", dest="reference_bits", action="store_true", default=None,
                    help="probabilities in CRFsuite's own operation order with a correctly rounded exp: the reference's output "
                         "files bit for bit (the default whenever the model has 2 labels and a window of <= 32 items)")
    ap.add_argument("--fast-kernels", dest="reference_bits", action="store_false",
                    help="the reorganised arithmetic of the fast kernels: probabilities within a few ulps of the reference's")
    ap.add_argument("--composition-domains", default=None,
                    help="file with one domain accession per line (the type classifier's domains.tsv): also write "
                         "<base>.compositions.npy, the classifier's input matrix")
    args = ap.parse_args(argv)
    crf = ClusterCRF.trained(args.model)
    crf.reference_bits = args.reference_bits  # None: ClusterCRF's default
    genes_t = tables.GeneTable.load(args.genes)
    feats_t = filter_features(tables.FeatureTable.load(args.features), args.e_filter, args.p_filter)
    comp_domains = None
    if args.composition_domains:
        with open(args.composition_domains) as fh:
            comp_domains = [line.strip() for line in fh if line.strip()]
    res = predict_tables(
        genes_t, feats_t, crf, pad=not args.no_pad, threshold=args.threshold, n_cds=args.cds,
        edge_distance=args.edge_distance, trim=not args.no_trim, composition_domains=comp_domains, criterion=args.postproc)
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
