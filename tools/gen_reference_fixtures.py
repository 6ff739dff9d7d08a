#!/usr/bin/env python3
"""BUILD CONTAINER ONLY: generate reference-produced golden vectors for the Python-level rows of SURVEY.md 8a.

Imports the REFERENCE'S OWN code from /root/reference (nothing of it is copied or shipped) and runs

* `gecco.crf.ClusterCRF.predict_probabilities`   (gecco/crf/__init__.py:148-273: sort, group, pad / skip + warnings, the sliding
  windows of gecco/_meta.py:124-132, max over windows, annotate (gecco/crf/features.py), cluster_weight, progress calls),
* `gecco.refine.ClusterRefiner.iter_clusters`     (gecco/refine.py:51-64,118-200: grouper, trim, both criteria, edge distance),
* `gecco.model.Cluster.domain_composition`        (gecco/model.py:458-503) and `Cluster.average_probability` (:442-447)

on seeded random inputs, writing DATA fixtures (inputs + the reference's outputs) to tests/golden/ref_*.json.

What is real and what is not.  The reference classes above are the reference's, imported as they are.  Two things the image
lacks are supplied here, in memory, and never reach the repository's product or the fixtures:

* `Bio` (Biopython) and `importlib_resources`: `gecco.model` imports names from them at module level; the rows exercised here
  only ever read `gene.source.id`.  Minimal placeholder modules are put into `sys.modules` for the import to succeed.
* `ClusterCRF.model` (a `sklearn_crfsuite.CRF` in the reference; [EXT] CRFsuite, absent from the image): an adapter with the
  two members the reference calls -- `predict_marginals_single(feats)` (answered by the repository's C oracle,
  oracle/crf_oracle.c, which restates CRFsuite's tagger and is pinned on the reference's own BGC0001866 fixture) and
  `state_features_` (oracle/lcrf.py's reading of the model file).  So the fixtures pin everything the REFERENCE'S PYTHON does
  around the tagger -- which is what rows D, X, W, N, R and f4 are -- while the tagger arithmetic itself stays pinned by
  tests/test_oracle_golden.py.

usage:  python tools/gen_reference_fixtures.py [--out tests/golden] [--cases 240]
"""
import argparse
import gzip
import json
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 0x6ECC0


# ---- placeholders for what the image lacks (in memory only) ------------------------------------------------------------------
def _install_placeholders():
    if "Bio" not in sys.modules:
        bio = types.ModuleType("Bio")
        bio.__version__ = "1.83"
        bio.__path__ = []  # (a package)
        seq = types.ModuleType("Bio.Seq")
        seqfeature = types.ModuleType("Bio.SeqFeature")
        seqrecord = types.ModuleType("Bio.SeqRecord")

        class Seq(str):
            pass

        class _Any:
            def __init__(self, *a, **k):
                self.args, self.kwargs = a, k

        class SeqRecord:
            def __init__(self, seq=None, id="<unknown id>", name="<unknown name>", **k):
                self.seq, self.id, self.name = seq, id, name

        seq.Seq = Seq
        for name in ("SeqFeature", "FeatureLocation", "CompoundLocation", "Reference"):
            setattr(seqfeature, name, type(name, (_Any,), {}))
        seqrecord.SeqRecord = SeqRecord
        bio.Seq, bio.SeqFeature, bio.SeqRecord = seq, seqfeature, seqrecord
        sys.modules.update({"Bio": bio, "Bio.Seq": seq, "Bio.SeqFeature": seqfeature, "Bio.SeqRecord": seqrecord})
    try:
        import importlib.resources.abc  # noqa: F401  (Python >= 3.11)
    except ImportError:
        if "importlib_resources" not in sys.modules:
            import importlib.abc
            import importlib.resources

            ir = types.ModuleType("importlib_resources")
            ir.__path__ = []
            ir.files = importlib.resources.files
            ir_abc = types.ModuleType("importlib_resources.abc")
            ir_abc.Traversable = importlib.abc.Traversable
            ir.abc = ir_abc
            sys.modules.update({"importlib_resources": ir, "importlib_resources.abc": ir_abc})


def import_reference():
    if not os.path.isdir(os.path.join(REFERENCE, "gecco")):
        raise SystemExit(f"{REFERENCE}/gecco not found: this generator runs in the build container only")
    _install_placeholders()
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, ROOT)
    import gecco.crf
    import gecco.model
    import gecco.refine

    assert os.path.abspath(gecco.crf.__file__).startswith(REFERENCE)
    return gecco


# ---- the tagger adapter ----------------------------------------------------------------------------------------------------
class OracleTagger:
    """The two members of `sklearn_crfsuite.CRF` the reference's inference path touches, answered by the C oracle."""

    def __init__(self, om):
        from oracle import lcrf

        self.om = om
        self.state_features_ = lcrf.state_features_view(om)
        self.calls = 0

    def predict_marginals_single(self, feats):
        from oracle import crf_oracle as orc

        self.calls += 1
        index = self.om["attr_index"]
        gptr, attr = [0], []
        for item in feats:
            attr.extend(index[name] for name in item if name in index)  # (names unknown to the model are dropped, as CRFsuite does)
            gptr.append(len(attr))
        cptr = np.array([0, len(feats)], dtype=np.int32)
        marg, _ = orc.full_marginals(self.om["state"], self.om["trans"], cptr, np.array(gptr, dtype=np.int32),
                                     np.array(attr if attr else [0], dtype=np.int32)[:max(len(attr), 1)] if attr else np.zeros(0, np.int32))
        labels = self.om["labels"]
        return [{labels[k]: float(row[k]) for k in range(len(labels))} for row in marg]


# ---- random inputs -----------------------------------------------------------------------------------------------------------
def random_genes(rng, gecco, om, contig_lengths, unknown_frac=0.1, empty_frac=0.3, shuffle=False, dup_frac=0.15, max_domains=4,
                 one_domain_max=False, tie_starts=False):
    """Reference `Gene` objects + their plain description.  Domain accessions: Zipf over the model's attributes, a share of
    names the model does not know, repeated accessions inside a gene (dict keys collapse them), domain lists out of start order."""
    model = gecco.model
    attrs = om["attrs"]
    hot = np.argsort(om["state"][:, 1] - om["state"][:, 0])[-150:]
    genes, desc = [], []
    for ci, n in enumerate(contig_lengths):
        src = sys.modules["Bio.SeqRecord"].SeqRecord(None, id=f"ctg{ci:03d}" if ci % 3 else f"Z_ctg{ci:03d}")
        pos = 1
        planted = rng.random() < 0.5 and n >= 6
        run0 = int(rng.integers(0, max(1, n - 5))) if planted else -1
        run1 = run0 + int(rng.integers(3, max(4, min(n, 25)))) if planted else -1
        for gi in range(n):
            start = pos + int(rng.integers(0, 300))
            if tie_starts and gi and rng.random() < 0.15:
                start = genes[-1].start  # (equal starts inside a contig: the sort is stable)
            end = start + int(rng.integers(90, 2400))
            pos = end + 1 if not tie_starts else start + 1
            k = 0 if rng.random() < empty_frac else 1 + min(int(rng.geometric(0.5)) - 1, max_domains - 1)
            if one_domain_max:
                k = min(k, 1)
            doms = []
            for di in range(k):
                if run0 <= gi < run1 and rng.random() < 0.8:
                    name = attrs[int(hot[int(rng.integers(0, len(hot)))])]
                elif rng.random() < unknown_frac:
                    name = f"PF9{int(rng.integers(0, 9999)):04d}x"
                else:
                    name = attrs[min(int(rng.zipf(1.2)) - 1, len(attrs) - 1)]
                if doms and rng.random() < dup_frac:
                    name = doms[int(rng.integers(0, len(doms)))].name
                ds = int(rng.integers(1, 400))
                doms.append(model.Domain(name, ds, ds + int(rng.integers(10, 200)), "Pfam", 1e-5, 1e-7))
            pid = f"{src.id}_{gi + 1}"
            gene = model.Gene(src, start, end, model.Strand.Coding if rng.random() < 0.5 else model.Strand.Reverse,
                              model.Protein(pid, None, doms))
            genes.append(gene)
    if shuffle:
        order = rng.permutation(len(genes))
        genes = [genes[i] for i in order]
    for g in genes:
        desc.append([g.source.id, g.protein.id, g.start, g.end, int(g.strand),
                     [[d.name, d.start, d.end] for d in g.protein.domains]])  # (hmm "Pfam", i_evalue 1e-5, pvalue 1e-7 throughout)
    return genes, desc


def describe_output(genes):
    """What the reference returned: gene order, the gene-level probability (None in domain mode for genes with domains, where the
    probabilities sit on the domains), and per domain -- in the order the reference left them -- name, cluster_weight and, when it
    is not the gene's, the domain's probability."""
    return {
        "order": [g.protein.id for g in genes],
        "p": [g._probability for g in genes],
        # (protein mode: a domain's probability IS its gene's -- checked here, stored once)
        "domains": [[[d.name, d.cluster_weight] if d.probability == g._probability else [d.name, d.cluster_weight, d.probability]
                     for d in g.protein.domains] for g in genes],
    }


# ---- predict_probabilities cases ---------------------------------------------------------------------------------------------
def gen_crf_cases(gecco, om, n_cases, rng):
    tagger = OracleTagger(om)
    cases = []
    grid = []
    for W in (5, 20, 32):
        for step in (1, 3, 20):
            if step > W:
                continue
            for pad in (True, False):
                grid.append((W, step, pad))
    k = 0
    while len(cases) < n_cases:
        W, step, pad = grid[k % len(grid)]
        mode = "domain" if k % 7 == 6 else "protein"
        k += 1
        nct = int(rng.integers(1, 6))
        lengths = []
        for _ in range(nct):
            r = rng.random()
            if r < 0.35:
                lengths.append(int(rng.integers(1, W)))  # shorter than the window: padded / skipped
            elif r < 0.5:
                lengths.append(W + int(rng.integers(-1, 2)))  # W - 1, W, W + 1
            else:
                lengths.append(int(rng.integers(W, 2 * W + 10)))
        genes, desc = random_genes(rng, gecco, om, lengths, shuffle=rng.random() < 0.5, unknown_frac=0.15 if k % 3 == 0 else 0.05,
                                   empty_frac=float(rng.choice([0.0, 0.3, 0.6])), one_domain_max=(mode == "domain" and rng.random() < 0.6),
                                   tie_starts=rng.random() < 0.2)
        crf = gecco.crf.ClusterCRF(feature_type=mode, window_size=W, window_step=step)
        crf.model = tagger
        calls = []
        case = {"params": {"feature_type": mode, "window_size": W, "window_step": step, "pad": pad}, "genes": desc}
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                out = crf.predict_probabilities(genes, pad=pad, progress=lambda i, t: calls.append((i, t)))
                case["expect"] = describe_output(out)
                # the caller's domain lists are sorted in place (crf/__init__.py:200-201): record the order the INPUT objects end in
                case["expect"]["input_sorted_in_place"] = all(
                    [d.start for d in g.protein.domains] == sorted(d.start for d in g.protein.domains) for g in genes)
            except Exception as err:  # the reference's own failure modes are part of its behaviour
                case["error"] = {"type": type(err).__name__, "message": str(err)[:300]}
        case["warnings"] = [[w.category.__name__, str(w.message)] for w in caught]
        if calls:
            totals = {t for _, t in calls}
            case["progress"] = {"n_calls": len(calls), "first": list(calls[0]), "last": list(calls[-1]), "total": calls[0][1],
                                "monotone": all(a[0] <= b[0] for a, b in zip(calls, calls[1:])), "one_total": len(totals) == 1}
        cases.append(case)
    return cases


# ---- refiner cases -----------------------------------------------------------------------------------------------------------
def gen_refiner_cases(gecco, om, n_cases, rng):
    model, refine = gecco.model, gecco.refine
    bio = sorted(refine.BIO_PFAMS)
    cases = []
    for k in range(n_cases):
        criterion = "antismash" if k % 4 == 3 else "gecco"
        params = {
            "threshold": float(rng.choice([0.8, 0.5, 0.3, 0.95])), "criterion": criterion,
            "n_cds": int(rng.choice([1, 2, 3, 5])), "n_biopfams": int(rng.choice([1, 2, 5])),
            "average_threshold": float(rng.choice([0.6, 0.3, 0.9])), "edge_distance": int(rng.integers(0, 4)),
            "trim": bool(rng.random() < 0.7),
        }
        nct = int(rng.integers(1, 5))
        # (the reference's antismash criterion fails on a cluster that holds a gene without probability: mostly avoided)
        none_frac = 0.12 if (criterion == "gecco" or k % 20 == 3) else 0.0
        genes, desc = [], []
        for ci in range(nct):
            src = sys.modules["Bio.SeqRecord"].SeqRecord(None, id=f"seq{ci}" if ci % 2 else f"Aseq{ci}")
            n = int(rng.integers(1, 60))
            pos = 1
            p = 0.1
            for gi in range(n):
                # a random walk with plateaus so that runs above the threshold exist
                if rng.random() < 0.15:
                    p = float(rng.choice([0.05, 0.2, 0.85, 0.99, params["threshold"], np.nextafter(params["threshold"], 1.0)]))
                prob = None if rng.random() < none_frac else float(np.clip(p + rng.normal(0, 0.03), 0.0, 1.0))
                if prob is not None and rng.random() < 0.05:
                    prob = params["threshold"]  # exactly the threshold: `>` is strict
                start = pos + int(rng.integers(0, 200))
                if gi and rng.random() < 0.08:
                    start = genes[-1].start  # equal starts: (start, end) decides
                end = start + int(rng.integers(90, 2000))
                pos = max(pos, end + 1)
                kdom = 0 if rng.random() < 0.35 else int(rng.integers(1, 4))
                doms = []
                for _ in range(kdom):
                    name = bio[int(rng.integers(0, len(bio)))] if (criterion == "antismash" and rng.random() < 0.5) else \
                        om["attrs"][int(rng.integers(0, len(om["attrs"])))]
                    ds = int(rng.integers(1, 300))
                    doms.append(model.Domain(name, ds, ds + 50, "Pfam", 1e-5, 1e-7, probability=prob))
                g = model.Gene(src, start, end, model.Strand.Coding, model.Protein(f"{src.id}_{gi + 1}", None, doms), _probability=prob)
                genes.append(g)
        if rng.random() < 0.4:
            order = rng.permutation(len(genes))
            genes = [genes[i] for i in order]
        desc = [[g.source.id, g.protein.id, g.start, g.end, g._probability, [d.name for d in g.protein.domains]] for g in genes]
        refiner = refine.ClusterRefiner(**params)
        # the CLI runs one refiner call per contig (cli/commands/_common.py:595-625); both ways are recorded
        case = {"params": params, "genes": desc}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (numpy's "mean of empty slice" for a cluster trimmed to nothing: the reference's own)
            try:
                case["clusters"] = [[c.id, [g.protein.id for g in c.genes], c.average_probability, c.maximum_probability, c.start, c.end]
                                    for c in refiner.iter_clusters(genes)]
            except Exception as err:  # (antismash criterion on genes without a probability: TypeError out of numpy.mean)
                case["error"] = {"type": type(err).__name__, "message": str(err)[:300]}
        cases.append(case)
    return cases


# ---- composition cases -------------------------------------------------------------------------------------------------------
def gen_composition_cases(gecco, om, n_cases, rng):
    model = gecco.model
    cases = []
    for k in range(n_cases):
        src = sys.modules["Bio.SeqRecord"].SeqRecord(None, id="c")
        names_pool = [om["attrs"][int(i)] for i in rng.integers(0, len(om["attrs"]), size=int(rng.integers(3, 40)))]
        genes, desc = [], []
        for gi in range(int(rng.integers(1, 30))):
            doms = []
            for _ in range(int(rng.integers(0, 5))):
                doms.append(model.Domain(names_pool[int(rng.integers(0, len(names_pool)))], 1, 50, "Pfam",
                                         float(10.0 ** -rng.uniform(0, 40)), float(10.0 ** -rng.uniform(0, 40))))
            genes.append(model.Gene(src, gi * 1000 + 1, gi * 1000 + 900, model.Strand.Coding, model.Protein(f"c_{gi + 1}", None, doms),
                                    _probability=float(rng.random())))
            desc.append([[d.name, d.i_evalue, d.pvalue] for d in doms])
        cluster = model.Cluster("c_cluster_1", genes)
        if k % 3 == 0:
            all_possible = None
        else:
            extra = [om["attrs"][int(i)] for i in rng.integers(0, len(om["attrs"]), size=int(rng.integers(0, 30)))]
            all_possible = sorted(set(names_pool[: len(names_pool) // 2] + extra))
        out = {}
        for normalize in (True, False):
            for pvalue in (True, False):
                vec = cluster.domain_composition(all_possible, normalize=normalize, minlog_weights=False, pvalue=pvalue)
                out[f"normalize={int(normalize)},pvalue={int(pvalue)}"] = [float(x) for x in vec]
        cases.append({"genes": desc, "all_possible": all_possible, "composition": out,
                      "gene_p": [g._probability for g in genes],
                      "average_probability": cluster.average_probability, "maximum_probability": cluster.maximum_probability})
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=GOLDEN)
    ap.add_argument("--cases", type=int, default=240)
    args = ap.parse_args()
    gecco = import_reference()
    from oracle import lcrf

    om = lcrf.load_model(os.path.join(GOLDEN, "model.pkl"), os.path.join(GOLDEN, "model.pkl.md5"))
    header = {
        "generator": "tools/gen_reference_fixtures.py (build container; imports /root/reference/gecco)",
        "reference_version": gecco.__version__,
        "tagger": "oracle/crf_oracle.c behind ClusterCRF.model (sklearn_crfsuite is absent from the image); weights: tests/golden/model.pkl",
        "seed": SEED,
    }
    sets = {
        "ref_predict_probabilities": gen_crf_cases(gecco, om, args.cases, np.random.default_rng(SEED)),
        "ref_refiner": gen_refiner_cases(gecco, om, args.cases, np.random.default_rng(SEED + 1)),
        "ref_composition": gen_composition_cases(gecco, om, max(60, args.cases // 3), np.random.default_rng(SEED + 2)),
    }
    for name, cases in sets.items():
        doc = dict(header, cases=cases)
        text = json.dumps(doc, separators=(",", ":"), allow_nan=False)
        path = os.path.join(args.out, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as fh:  # (mtime 0: the same bytes from the same inputs)
            fh.write(text.encode())
        n_err = sum(1 for c in cases if "error" in c)
        print(f"{path}: {len(cases)} cases ({n_err} where the reference raises), {len(text) / 1e6:.2f} MB of JSON, {os.path.getsize(path) / 1e3:.0f} KB on disk")


if __name__ == "__main__":
    main()
