"""Drop-in ``ClusterCRF`` for GECCO's ``crf_type=`` injection point, backed by the HIP engine.

Mirror of ``/root/reference/gecco/crf/__init__.py:55-273`` (the inference half): same class
surface -- ``trained`` / ``__init__`` / ``predict_probabilities`` / ``fit`` / ``save`` and the
attributes ``feature_type window_size window_step algorithm significance
significant_features model`` -- same argument meaning, same warnings and errors.  GECCO hands
a *class* down its CLI (``gecco/cli/commands/__init__.py:127-137,160-163``) and calls
``crf_type.trained(model)`` then ``.predict_probabilities(genes, pad=, progress=)``
(``gecco/cli/commands/_common.py:577,588-592``); use it as::

    import gecco.cli, gecco_amd.crf
    gecco.cli.main(crf_type=gecco_amd.crf.ClusterCRF)

What differs is where the arithmetic runs: instead of one python-crfsuite call per sliding
window (``:253``), every contig of the call is packed into one CSR batch and scored by the
HIP kernels through the C ABI (``include/gecco_crf.h``).  There is no CPU fallback.
"""
import gc
import itertools
import operator
import os
import warnings
from typing import Any, Callable, Dict, FrozenSet, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native, packing, pickle_model
from ._objpath_loader import module as _objpath_module

__all__ = ["ClusterCRF", "NotFittedError"]

try:  # the reference documents sklearn's NotFittedError (crf/__init__.py:176)
    from sklearn.exceptions import NotFittedError
except Exception:  # pragma: no cover

    class NotFittedError(ValueError, AttributeError):  # type: ignore
        pass


class _CRFSuiteModelView:
    """What ``ClusterCRF.model`` exposes of a fitted ``sklearn_crfsuite.CRF`` [EXT], rebuilt
    from the CRFsuite model blob: ``classes_``, ``attributes_``, ``state_features_``,
    ``transition_features_``, ``predict_marginals_single`` / ``predict_single``.
    Hyper-parameters of the pickled estimator (``c1``, ``c2`` ...) are kept as attributes."""

    def __init__(self, native_model: "_native.Model", params: Optional[Dict[str, Any]] = None,
                 training_log: Any = None):
        self.native = native_model
        self.training_log_ = training_log
        for k, v in (params or {}).items():
            if not hasattr(self, k):
                setattr(self, k, v)
        self.classes_: List[str] = native_model.labels()
        self.attributes_: List[str] = native_model.attrs()
        self._attr_index: Dict[str, int] = {a: i for i, a in enumerate(self.attributes_)}
        self._state: Optional[Dict[Tuple[str, str], float]] = None
        self._trans: Optional[Dict[Tuple[str, str], float]] = None
        self._w1: Optional[Dict[str, float]] = None

    # [EXT] sklearn-crfsuite parses these back from CRFsuite's text dump, which prints weights
    # with "%f": values are rounded to 6 decimals there, and so they are here.
    @property
    def state_features_(self) -> Dict[Tuple[str, str], float]:
        if self._state is None:
            w, present = self.native.state_weights()
            a_idx, l_idx = np.nonzero(present)
            self._state = {
                (self.attributes_[a], self.classes_[l]): float("%f" % w[a, l]) for a, l in zip(a_idx.tolist(), l_idx.tolist())
            }
        return self._state

    @property
    def transition_features_(self) -> Dict[Tuple[str, str], float]:
        if self._trans is None:
            w, present = self.native.trans_weights()
            self._trans = {
                (self.classes_[i], self.classes_[j]): float("%f" % w[i, j])
                for i in range(len(self.classes_)) for j in range(len(self.classes_)) if present[i, j]
            }
        return self._trans

    @property
    def cluster_weights_(self) -> Dict[str, float]:
        """``{domain: state_features_[(domain, '1')]}``: what `predict_probabilities` annotates domains with
        (crf/__init__.py:261-269).  Built once per model: filtering the 4 211 state features on every call cost more than
        scoring a 50-gene contig."""
        if self._w1 is None:
            self._w1 = {name: w for (name, lab), w in self.state_features_.items() if lab == "1"}
        return self._w1

    @property
    def num_attributes_(self) -> int:
        return self.native.num_attrs

    def _pack_single(self, xseq: Sequence[Dict[str, Any]]):
        gene_ptr = [0]
        attr: List[int] = []
        for item in xseq:
            for name, value in item.items():
                if value is not True and value != 1:
                    raise ValueError("only boolean/unit feature values are supported (GECCO emits {name: True})")
                idx = self._attr_index.get(name)
                if idx is not None:
                    attr.append(idx)
            gene_ptr.append(len(attr))
        return np.array([0, len(xseq)], dtype=np.int32), np.array(gene_ptr, dtype=np.int32), np.array(attr, dtype=np.int32)

    def predict_marginals_single(self, xseq: Sequence[Dict[str, Any]], device: int = 0) -> List[Dict[str, float]]:
        """[EXT] ``CRF.predict_marginals_single``: whole-sequence marginals of every label."""
        if len(xseq) == 0:
            return []
        cptr, gptr, attr = self._pack_single(xseq)
        marg, _ = self.native.marginals_full(cptr, gptr, attr, device=device)
        return [dict(zip(self.classes_, row.tolist())) for row in marg]

    def predict_single(self, xseq: Sequence[Dict[str, Any]], device: int = 0) -> List[str]:
        """[EXT] ``CRF.predict_single``: Viterbi labels."""
        if len(xseq) == 0:
            return []
        cptr, gptr, attr = self._pack_single(xseq)
        y, _ = self.native.viterbi(cptr, gptr, attr, device=device)
        return [self.classes_[i] for i in y.tolist()]


_FAST_CLONE: Dict[type, bool] = {}
_IS_DATACLASS: Dict[type, bool] = {}


def _is_dataclass(obj: Any) -> bool:
    cls = type(obj)
    r = _IS_DATACLASS.get(cls)
    if r is None:
        import dataclasses

        r = _IS_DATACLASS[cls] = dataclasses.is_dataclass(cls)
    return r


def _replace(obj: Any, changes: Dict[str, Any]) -> Any:
    """``dataclasses.replace`` for the model's frozen dataclasses.  ``replace`` re-runs ``__init__`` field by
    field (~6 us per object: 90 % of predict_probabilities on a metagenome); a dataclass without ``__slots__``
    and without ``__post_init__`` is fully described by its ``__dict__``, so the copy is made there."""
    cls = type(obj)
    fast = _FAST_CLONE.get(cls)
    if fast is None:
        fast = _FAST_CLONE[cls] = (hasattr(obj, "__dict__") and not hasattr(cls, "__post_init__")
                                   and not hasattr(cls, "__slots__"))
    if not fast:
        import dataclasses

        return dataclasses.replace(obj, **changes)
    new = object.__new__(cls)
    d = new.__dict__
    d.update(obj.__dict__)
    d.update(changes)
    return new


def _annotate(gene: Any, gene_p: Optional[float], domain_p: Optional[List[float]],
              weights: Dict[Tuple[str, str], float]) -> Any:
    """New gene carrying the probabilities and the domains' cluster weights.

    Equivalent to the reference's chain ``gene.with_probability(p)`` (``model.py:364-375``; or
    per-domain ``with_probability`` in domain mode, ``features.py:109-117``) followed by
    ``with_protein(protein.with_domains(d.with_cluster_weight(w) ...))`` (``crf/__init__.py:261-269``),
    but for dataclass models (GECCO's and this package's) each object is rebuilt once instead of
    twice; any other duck-typed model goes through its own ``with_*`` methods."""
    doms = gene.protein.domains
    if _is_dataclass(gene) and all(_is_dataclass(d) for d in doms):
        new_doms = []
        for j, d in enumerate(doms):
            changes: Dict[str, Any] = {"cluster_weight": weights.get((d.name, "1")), "qualifiers": d.qualifiers.copy()}
            if gene_p is not None:
                changes["probability"] = gene_p
            elif domain_p is not None:
                changes["probability"] = domain_p[j]
            new_doms.append(_replace(d, changes))
        protein = _replace(gene.protein, {"domains": new_doms})
        gchanges: Dict[str, Any] = {"protein": protein, "qualifiers": gene.qualifiers.copy()}
        if gene_p is not None:
            gchanges["_probability"] = gene_p
        return _replace(gene, gchanges)
    if gene_p is not None:
        gene = gene.with_probability(gene_p)
    elif domain_p is not None:
        gene = gene.with_protein(gene.protein.with_domains(
            [d.with_probability(p) for d, p in zip(gene.protein.domains, domain_p)]))
    return gene.with_protein(gene.protein.with_domains(
        d.with_cluster_weight(weights.get((d.name, "1"))) for d in gene.protein.domains))


_PLAIN_MODELS: Dict[Any, bool] = {}  # (Gene, Protein, Domain classes) -> plain dataclasses with GECCO's field names?


def _annotate_all(genes: List[Any], probs: List[float], w1: Dict[str, float]) -> Optional[List[Any]]:
    """`_annotate(gene, p, None, weights)` for a whole list of genes of ONE plain dataclass model (GECCO's, or this
    package's): the per-object work is two dict operations and nothing is looked up twice.  Returns None when the
    objects are of any other kind (the caller then goes gene by gene).  On a metagenome this loop IS
    predict_probabilities: rebuilding ~3.5 objects per gene is all the host still does."""
    if not genes:
        return []
    g0 = genes[0]
    gene_cls, prot_cls = type(g0), type(g0.protein)
    dom_cls = None
    for g in genes:
        if g.protein.domains:
            dom_cls = type(g.protein.domains[0])
            break
    # (the shape checks below look at classes and at the field names of one object of each: remembered per class triple --
    # on a contig of a few dozen genes they cost as much as the copies)
    known = _PLAIN_MODELS.get((gene_cls, prot_cls, dom_cls))
    if known is not None:
        if not known:
            return None
        native = _objpath_module()
        if native is not None:
            return native.annotate_all(genes, probs, w1, gene_cls, prot_cls, dom_cls)
    key = (gene_cls, prot_cls, dom_cls)
    for obj, cls in ((g0, gene_cls), (g0.protein, prot_cls)):
        if not _is_dataclass(obj) or hasattr(cls, "__post_init__") or hasattr(cls, "__slots__") or not hasattr(obj, "__dict__"):
            _PLAIN_MODELS[key] = False
            return None
    if dom_cls is not None and (not _IS_DATACLASS.setdefault(dom_cls, __import__("dataclasses").is_dataclass(dom_cls))
                                or hasattr(dom_cls, "__post_init__") or hasattr(dom_cls, "__slots__")):
        _PLAIN_MODELS[key] = False
        return None
    # ... with GECCO's field names (gecco/model.py:274-290,321-375): anything else goes through its own with_* methods
    if not {"protein", "qualifiers", "_probability"} <= g0.__dict__.keys() or "domains" not in g0.protein.__dict__:
        _PLAIN_MODELS[key] = False
        return None
    if dom_cls is not None:
        d0 = next(g for g in genes if g.protein.domains).protein.domains[0]
        if not hasattr(d0, "__dict__") or not {"name", "probability", "cluster_weight", "qualifiers"} <= d0.__dict__.keys():
            _PLAIN_MODELS[key] = False
            return None
    if dom_cls is not None:  # (a batch without a single domain says nothing about the domain class: not remembered)
        _PLAIN_MODELS[key] = True
    from ._objpath_loader import module

    native = module()  # csrc/objpath.c: the loop below against the CPython C API (tp_alloc + PyDict_Copy per object)
    if native is not None:
        return native.annotate_all(genes, probs, w1, gene_cls, prot_cls, dom_cls)
    if not isinstance(probs, list):
        probs = np.asarray(probs).tolist()
    wget = w1.get  # domain name -> weight of its ('name', '1') state feature
    new = object.__new__
    out = []
    append = out.append
    for gene, p in zip(genes, probs):
        gd = gene.__dict__
        prot = gd["protein"]
        if type(gene) is not gene_cls or type(prot) is not prot_cls:
            return None
        pd = prot.__dict__
        new_doms = []
        for d in pd["domains"]:
            if type(d) is not dom_cls:
                return None
            nd = new(dom_cls)
            dd = nd.__dict__
            dd.update(d.__dict__)
            dd["probability"] = p
            dd["cluster_weight"] = wget(dd["name"])
            dd["qualifiers"] = dd["qualifiers"].copy()
            new_doms.append(nd)
        np_ = new(prot_cls)
        npd = np_.__dict__
        npd.update(pd)
        npd["domains"] = new_doms
        ng = new(gene_cls)
        ngd = ng.__dict__
        ngd.update(gd)
        ngd["protein"] = np_
        ngd["qualifiers"] = gd["qualifiers"].copy()
        ngd["_probability"] = p
        append(ng)
    return out


def _default_devices() -> List[int]:
    env = os.environ.get("GECCO_HIP_DEVICES", "").strip()
    if env:
        return [int(x) for x in env.split(",") if x.strip() != ""]
    return [0]


class ClusterCRF(object):
    """A GECCO-compatible CRF whose inference runs on MI355X."""

    _FILENAME = pickle_model.MODEL_FILENAME
    #: contigs are handed to the batch driver in calls of at most this many genes, `progress` is called after each (the
    #: driver cuts a call into chunks itself and overlaps their copies and kernels: a 2 M-gene metagenome is ONE call)
    _BATCH_GENES = 1 << 24

    # ------------------------------------------------------------------ construction
    @classmethod
    def trained(cls, model_path: Union[str, os.PathLike, Any, None] = None) -> "ClusterCRF":
        """Load a pre-trained model directory (``model.pkl`` + ``model.pkl.md5``).

        Must be overridden rather than inherited: the pickle hard-codes
        ``gecco.crf.ClusterCRF``.  Raises ``ValueError("MD5 hash of model data does not match
        signature")`` like the reference (``gecco/crf/__init__.py:96-97``).
        """
        if model_path is None:
            model_path = cls._embedded_model_dir()
        record = pickle_model.load_model_dir(model_path)
        st = record.state
        self = cls.__new__(cls)
        self.feature_type = st.get("feature_type", "protein")
        self.window_size = st.get("window_size", 5)
        self.window_step = st.get("window_step", 1)
        self.algorithm = st.get("algorithm", "lbfgs")
        self.significance = st.get("significance")
        self.significant_features = st.get("significant_features")
        self._options = st.get("_options", {"algorithm": self.algorithm})
        self._record = record
        self.devices = _default_devices()
        self.reference_bits: Optional[bool] = None  # None: on when the mode covers the model (_reference_bits_now)
        crf_state = st["model"].state if st.get("model") is not None else None
        if crf_state is None:
            self.model = None
        else:
            params = {k: v for k, v in crf_state.items() if k not in ("modelfile", "training_log_", "_tagger", "_info_cached")}
            native = _native.Model.from_lcrf(pickle_model.crfsuite_blob(record))
            self.model = _CRFSuiteModelView(native, params, crf_state.get("training_log_"))
        return self

    @staticmethod
    def _embedded_model_dir():
        """Directory of the embedded model: GECCO's own package data when GECCO is installed
        (``files("gecco.crf")``, crf/__init__.py:78), else $GECCO_AMD_MODEL_DIR."""
        env = os.environ.get("GECCO_AMD_MODEL_DIR")
        if env:
            return env
        try:
            from importlib.resources import files
            import importlib.util

            spec = importlib.util.find_spec("gecco")
            if spec is not None and spec.submodule_search_locations:
                import pathlib

                for loc in spec.submodule_search_locations:
                    cand = pathlib.Path(loc) / "crf"
                    if (cand / "model.pkl").exists():
                        return cand
            return files("gecco.crf")
        except Exception as err:
            raise FileNotFoundError(
                "no embedded model: GECCO is not installed; pass a model directory or set GECCO_AMD_MODEL_DIR"
            ) from err

    def __init__(self, feature_type: str = "protein", algorithm: str = "lbfgs", window_size: int = 5,
                 window_step: int = 1, **kwargs: Any) -> None:
        # same checks, same messages as gecco/crf/__init__.py:132-137
        if feature_type not in {"protein", "domain"}:
            raise ValueError(f"invalid feature type: {feature_type!r}")
        if window_size <= 0:
            raise ValueError("Window size must be strictly positive")
        if window_step <= 0 or window_step > window_size:
            raise ValueError("Window step must be strictly positive and under `window_size`")
        self.feature_type = feature_type
        self.window_size = window_size
        self.window_step = window_step
        self.algorithm = algorithm
        self.significance: Optional[Dict[str, float]] = None
        self.significant_features: Optional[FrozenSet[str]] = None
        self.model: Optional[_CRFSuiteModelView] = None
        self._options = {"algorithm": algorithm, **kwargs}
        self._record = None
        self.devices = _default_devices()
        self.reference_bits: Optional[bool] = None  # None: on when the mode covers the model (_reference_bits_now)

    # ------------------------------------------------------------------ inference
    def predict_probabilities(self, genes: Iterable[Any], *, pad: bool = True,
                              progress: Optional[Callable[[int, int], None]] = None) -> List[Any]:
        """Predict how likely each gene is to be part of a gene cluster.

        Same contract as ``gecco/crf/__init__.py:148-273``: genes are sorted by
        ``(source.id, start)`` and their domains by ``start`` (in place, like the reference),
        grouped by contig; contigs shorter than the window are centre-padded (``pad=True``,
        with the reference's warning) or skipped; every gene gets the maximum, over the
        sliding windows covering it, of the window-local marginal P(label '1'); domains get
        ``cluster_weight`` from the model's state features; new ``Gene`` objects are returned
        in sorted order.
        """
        _progress = progress or (lambda x, y: None)
        if self.model is None:
            raise NotFittedError("This ClusterCRF instance is not fitted yet.")
        if self.feature_type not in ("protein", "domain"):
            raise ValueError(f"invalid feature type: {self.feature_type!r}")

        # :199-206 -- sort (mutating the caller's domain lists, as the reference does), group
        contigs: Optional[List[List[Any]]] = None
        batch: Optional[packing.PackedBatch] = None
        if self.feature_type == "protein":
            # csrc/objpath.c: ONE pass that checks the order, sorts the domain lists that need it, groups and packs the
            # features (:209-214) -- every object is visited once, while it is in cache
            native = _objpath_module()
            if native is not None:
                genes = genes if isinstance(genes, (list, tuple)) else list(genes)
                got = native.sort_group(genes, operator.attrgetter("start"), self.model._attr_index)
                if got is not None:
                    genes, contigs, ip, ap, at = got
                    batch = packing.PackedBatch(np.frombuffer(ip, dtype=np.int64), np.frombuffer(ap, dtype=np.int64),
                                                np.frombuffer(at, dtype=np.int32))
        if contigs is None:
            genes = sorted(genes, key=operator.attrgetter("source.id", "start"))
            for gene in genes:
                gene.protein.domains.sort(key=operator.attrgetter("start"))
            contigs = [list(g) for _, g in itertools.groupby(genes, key=operator.attrgetter("source.id"))]

        # :209-236 -- features -> CSR items; decide padding / skipping per contig
        W, step = self.window_size, self.window_step
        if batch is None:
            batch = packing.pack_contigs(contigs, self.model._attr_index, self.feature_type)
        scored = np.ones(len(contigs), dtype=bool)
        total = 0
        items_per_contig = np.diff(batch.item_ptr)
        for ci in np.flatnonzero(items_per_contig < W).tolist():  # (the warnings, in contig order)
            contig = contigs[ci]
            n_items = int(items_per_contig[ci])
            if pad:
                unit = self.feature_type if W - n_items == 1 else f"{self.feature_type}s"
                warnings.warn(
                    f"Contig {contig[0].source.id!r} does not contain enough"
                    f" {self.feature_type}s ({len(contig)}) for sliding window"
                    f" of size {W}, padding with"
                    f" {W - n_items} {unit}"
                )
            else:
                warnings.warn(
                    f"Contig {contig[0].source.id!r} does not contain enough"
                    f" {self.feature_type}s ({len(contig)}) for sliding window"
                    f" of size {W}"
                )
                scored[ci] = False
        total = int((np.maximum(items_per_contig[scored], W) - W + 1).sum())  # :239
        _progress(0, total)

        # :244-258 -- windowed marginals of label '1', batched over contigs
        label = self.model.native.label_id("1")
        if label < 0:
            raise ValueError("the model has no label '1'")
        p_items = self._score(batch, W, step, label, pad, _progress, total)

        # :258, features.py:74-120 (probabilities) and :261-269 (cluster weight = state feature
        # weight of (domain, '1'), None if absent) -- new Gene/Protein/Domain objects; contigs that
        # were skipped keep their probabilities and only get the weights.
        weights = self.model.state_features_
        w1 = self.model.cluster_weights_
        predicted: List[Any] = []
        # Millions of small container objects are about to be allocated and none of them dies: the cyclic collector's
        # generational passes over the growing heap cost 6x the allocations themselves (measured: 430 -> 63 ms per 50 000
        # genes).  Collection is suspended for the loop and restored to what the caller had.
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._annotate_contigs(contigs, scored, batch, p_items, weights, w1, predicted,
                                   genes if isinstance(genes, list) else None)
        finally:
            if gc_was_enabled:
                gc.enable()
        return predicted

    def _annotate_contigs(self, contigs, scored, batch, p_items, weights, w1, predicted, genes=None) -> None:
        if genes is not None and self.feature_type == "protein" and len(genes) == len(p_items) and bool(scored.all()):
            fast = _annotate_all(genes, np.ascontiguousarray(p_items, dtype=np.float64), w1)  # one native pass over all contigs
            if fast is not None:
                predicted.extend(fast)
                return
        for ci, contig in enumerate(contigs):
            if not scored[ci]:
                predicted.extend(_annotate(gene, None, None, weights) for gene in contig)
                continue
            i0 = int(batch.item_ptr[ci])
            if self.feature_type == "protein":
                probs = p_items[i0:i0 + len(contig)].tolist()
                fast = _annotate_all(contig, probs, w1)
                if fast is not None:
                    predicted.extend(fast)
                else:
                    for gene, p in zip(contig, probs):
                        predicted.append(_annotate(gene, p, None, weights))
            else:
                k = i0
                for gene in contig:
                    doms = gene.protein.domains
                    if doms:
                        predicted.append(_annotate(gene, None, [float(x) for x in p_items[k:k + len(doms)]], weights))
                        k += len(doms)
                    else:
                        predicted.append(_annotate(gene, float(p_items[k]), None, weights))
                        k += 1

    def predict_clusters(self, genes: Iterable[Any], *, pad: bool = True, threshold: float = 0.8,
                         criterion: str = "gecco", n_cds: int = 3, edge_distance: int = 0, trim: bool = True,
                         progress: Optional[Callable[[int, int], None]] = None) -> List[Any]:
        """Probabilities + cluster calls in one go: ``predict_probabilities`` followed by
        ``ClusterRefiner.iter_clusters`` with the CLI's defaults (threshold 0.8, cds 3, edge
        distance 0, trim; ``gecco/cli/commands/_parser.py:277-335``, ``_common.py:595-625``).
        Not a method of the reference class: a convenience of this one."""
        from .refine import ClusterRefiner

        annotated = self.predict_probabilities(genes, pad=pad, progress=progress)
        refiner = ClusterRefiner(threshold=threshold, criterion=criterion, n_cds=n_cds,
                                 edge_distance=edge_distance, trim=trim)
        clusters: List[Any] = []
        for _, group in itertools.groupby(annotated, key=lambda g: g.source.id):  # per contig, like the CLI
            clusters.extend(refiner.iter_clusters(list(group)))
        return clusters

    def predict_probabilities_csr(self, contig_ptr, gene_ptr, attr_id, *, pad: bool = True, label: str = "1",
                                  device: Optional[int] = None) -> np.ndarray:
        """Columnar entry point: the same scores for an already packed CSR batch (ids from
        ``self.model.attributes_``); returns one float64 per gene (NaN = contig skipped)."""
        if self.model is None:
            raise NotFittedError("This ClusterCRF instance is not fitted yet.")
        lab = self.model.native.label_id(label)
        if device is not None and [device] != list(self.devices):
            # a session of its own on the named device, in this object's mode: the bits do not depend on `device`
            ses = _native.Session(self.model.native, [device])
            ses.set_reference_bits(self._reference_bits_now())
            return ses.windowed_marginals(contig_ptr, gene_ptr, attr_id, self.window_size, self.window_step, lab, pad)
        return self._session().windowed_marginals(contig_ptr, gene_ptr, attr_id, self.window_size, self.window_step, lab, pad)

    def _session(self) -> "_native.Session":
        """The batch driver bound to ``self.devices`` (``gecco_crf_session_*``): created on first use, kept
        for the life of the object so that streams, plans and device buffers are reused from call to call."""
        devices = tuple(self.devices or [0])
        native = self.model.native
        ses = getattr(self, "_ses", None)
        # the session is bound to ONE native model: `fit`, `crf.model = ...` or a copy followed by a model swap must
        # not keep scoring with the previous model's weights
        if ses is None or ses[0] != devices or ses[1].model is not native:
            ses = (devices, _native.Session(native, devices))
            self._ses = ses
        ses[1].set_reference_bits(self._reference_bits_now())
        return ses[1]

    def _reference_bits_now(self) -> bool:
        """Whether this object's calls run in reference-bits mode (csrc/crf_exact.hip: CRFsuite's own operation order with a
        CORRECTLY ROUNDED exp).  Proven: genes.tsv / features.tsv / clusters.tsv of the reference's BGC0001866 fixture come out
        string-identical (its acceptance test compares whole files, /root/reference/galaxy/gecco.xml:83-111), and every gene of
        the 2 M-gene benchmark batch carries the bits of the oracle run with a correctly rounded exp.  Not claimed: the bits of
        CRFsuite on any host -- it calls the host's libm exp, which is not correctly rounded on every argument (against the
        oracle with this box's glibc exp 0.14 % of the genes differ, by <= 14 ulps; the fast kernels differ on 89 %, by <= 64).
        ``reference_bits`` True / False decides; None (the default) means: ``GECCO_AMD_REFERENCE_BITS=0|1`` if set, else ON
        whenever the mode covers the model (2 labels, window <= 32 items): every caller of this class is bound by its object
        or table handling (0.8 / 30 M genes/s), not by the kernels, and 99.86 % of the genes then carry the libm oracle's
        bits instead of 11 %.  The C ABI's default stays the fast kernels."""
        want = getattr(self, "reference_bits", None)
        if want is None:
            env = os.environ.get("GECCO_AMD_REFERENCE_BITS")
            if env in ("0", "1"):
                want = env == "1"
        covered = self.model is not None and self.model.native.num_labels == 2 and 1 <= int(self.window_size) <= 32
        return covered if want is None else bool(want)

    def _score(self, batch: "packing.PackedBatch", W: int, step: int, label: int, pad: bool,
               progress: Callable[[int, int], None], total: int) -> np.ndarray:
        """Run the batch through the HIP engine.  The native batch driver cuts it into chunks at contig
        boundaries, deals them to ``self.devices`` longest-first by gene count (independent queues, no
        collective) and pipelines upload / kernels / download on every device; here the batch is only
        split into pieces of <= _BATCH_GENES items so that `progress` is called in between."""
        n_contigs = len(batch.item_ptr) - 1
        out = np.full(int(batch.item_ptr[-1]), np.nan, dtype=np.float64)
        if n_contigs == 0:
            return out
        session = self._session()
        done = 0
        c0 = 0
        while c0 < n_contigs:
            c1 = c0 + 1
            while c1 < n_contigs and batch.item_ptr[c1 + 1] - batch.item_ptr[c0] <= self._BATCH_GENES:
                c1 += 1
            i0, i1 = int(batch.item_ptr[c0]), int(batch.item_ptr[c1])
            a0, a1 = int(batch.attr_ptr[i0]), int(batch.attr_ptr[i1])
            if c0 == 0 and c1 == n_contigs:
                cptr, gptr = batch.item_ptr, batch.attr_ptr
            else:
                cptr, gptr = batch.item_ptr[c0:c1 + 1] - i0, batch.attr_ptr[i0:i1 + 1] - a0
            out[i0:i1] = session.windowed_marginals(cptr, gptr, batch.attr_id[a0:a1], W, step, label, pad)
            n_items = np.diff(batch.item_ptr[c0:c1 + 1])
            scored = n_items >= W if not pad else np.ones(len(n_items), dtype=bool)
            done += int((np.maximum(n_items[scored], W) - W + 1).sum())
            progress(done, total)
            c0 = c1
        return out

    # ------------------------------------------------------------------ training (delegated)
    def fit(self, genes: Iterable[Any], **kwargs: Any) -> None:
        """Training is outside this engine's scope: delegate to the reference
        implementation when sklearn-crfsuite is importable (``gecco/crf/__init__.py:275-378``)."""
        try:
            from gecco.crf import ClusterCRF as _Reference  # type: ignore
        except Exception as err:
            raise NotImplementedError(
                "ClusterCRF.fit needs GECCO with sklearn-crfsuite (L-BFGS training runs in CRFsuite)"
            ) from err
        ref = _Reference(self.feature_type, self.algorithm, self.window_size, self.window_step,
                         **{k: v for k, v in self._options.items() if k != "algorithm"})
        ref.fit(genes, **kwargs)
        import tempfile

        with tempfile.TemporaryDirectory() as tmp:
            ref.save(tmp)
            fitted = type(self).trained(tmp)
        self.__dict__.pop("_ses", None)  # bound to the previous model
        self.__dict__.update(fitted.__dict__)

    def save(self, model_path: Union[str, os.PathLike]) -> None:
        """Write ``model.pkl`` (protocol 4) + ``model.pkl.md5`` loadable by stock GECCO
        (``gecco/crf/__init__.py:380-402``)."""
        if self._record is None:
            raise NotFittedError("This ClusterCRF instance is not fitted yet.")
        st = self._record.state
        st.update(feature_type=self.feature_type, window_size=self.window_size, window_step=self.window_step,
                  algorithm=self.algorithm, significance=self.significance,
                  significant_features=self.significant_features)
        pickle_model.dump_model_dir(self._record, model_path)
