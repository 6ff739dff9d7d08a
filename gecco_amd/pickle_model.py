"""Reader (and re-writer) of the ``model.pkl`` files GECCO ships and ``gecco train`` produces.

Mirrors ``ClusterCRF.trained`` (``/root/reference/gecco/crf/__init__.py:61-99``): stream-md5
the pickle against ``model.pkl.md5`` (case-insensitive), raise
``ValueError("MD5 hash of model data does not match signature")`` on mismatch, then unpickle.
The pickle names ``gecco.crf.ClusterCRF`` -> ``sklearn_crfsuite.estimator.CRF`` ->
``sklearn_crfsuite._fileresource.FileResource`` (+ ``pycrfsuite._logparser.TrainLogParser``);
none of those packages is needed here: a restricted unpickler maps exactly these four globals
to inert record classes (anything else is refused), and the CRFsuite model bytes are taken
from ``FileResource.__dict__['__FILE_RESOURCE_DATA__']`` in memory (the reference writes them
to a temp file for ``pycrfsuite.Tagger.open`` [EXT]).
"""
import functools
import hashlib
import io
import pathlib
import pickle
from typing import Any, BinaryIO, Dict, Tuple, Union

MODEL_FILENAME = "model.pkl"

_ALLOWED_GLOBALS = {
    ("gecco.crf", "ClusterCRF"),
    ("sklearn_crfsuite.estimator", "CRF"),
    ("sklearn_crfsuite._fileresource", "FileResource"),
    ("pycrfsuite._logparser", "TrainLogParser"),
}
_ALLOWED_BUILTINS = {"frozenset", "set", "dict", "list", "tuple", "bytes", "bytearray", "float", "int", "str", "bool"}


class PickledRecord:
    """Inert stand-in for an instance of one of the allow-listed classes."""

    _global: Tuple[str, str] = ("", "")

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        self.state: Dict[str, Any] = {}

    def __setstate__(self, state: Any) -> None:
        self.__dict__["state"] = dict(state) if isinstance(state, dict) else {"__state__": state}

    def __reduce_ex__(self, protocol: int):  # used by `dump_model`
        return (_RecordFactory(self._global), (), self.state)


_record_types: Dict[Tuple[str, str], type] = {}


def _record_type(module: str, name: str) -> type:
    key = (module, name)
    if key not in _record_types:
        _record_types[key] = type(name, (PickledRecord,), {"_global": key})
    return _record_types[key]


class _RecordFactory:
    """Callable pickled *by reference to the original class path* (see `_CompatPickler`)."""

    def __init__(self, glob: Tuple[str, str]):
        self.glob = glob

    def __call__(self):  # pragma: no cover - only meaningful inside the reference
        return _record_type(*self.glob)()


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if (module, name) in _ALLOWED_GLOBALS:
            return _record_type(module, name)
        if module == "builtins" and name in _ALLOWED_BUILTINS:
            import builtins

            return getattr(builtins, name)
        raise pickle.UnpicklingError(f"refusing to load global {module}.{name} from a GECCO model pickle")


def _md5_stream(fh: BinaryIO) -> str:
    hasher = hashlib.md5()
    read = functools.partial(fh.read, io.DEFAULT_BUFFER_SIZE)
    for chunk in iter(read, b""):
        hasher.update(chunk)
    return hasher.hexdigest()


def load_model_dir(model_path: Union[str, "pathlib.Path", Any]) -> PickledRecord:
    """md5-check and unpickle ``<model_path>/model.pkl``; `model_path` may be a directory path
    or an ``importlib.resources`` Traversable (anything with ``joinpath(...).open``)."""
    if not hasattr(model_path, "joinpath"):
        model_path = pathlib.Path(model_path)
    with model_path.joinpath(f"{MODEL_FILENAME}.md5").open() as sig:
        signature = sig.read().strip()
    with model_path.joinpath(MODEL_FILENAME).open("rb") as fh:
        if _md5_stream(fh).upper() != signature.upper():
            raise ValueError("MD5 hash of model data does not match signature")
        fh.seek(0)
        obj = RestrictedUnpickler(fh).load()
    if not isinstance(obj, PickledRecord) or obj._global != ("gecco.crf", "ClusterCRF"):
        raise ValueError("model.pkl does not contain a gecco.crf.ClusterCRF object")
    return obj


def crfsuite_blob(record: PickledRecord) -> bytes:
    """The raw CRFsuite model bytes inside an unpickled ClusterCRF record."""
    crf = record.state.get("model")
    if crf is None:
        raise ValueError("the pickled ClusterCRF has no fitted model")
    res = crf.state.get("modelfile")
    blob = None if res is None else res.state.get("__FILE_RESOURCE_DATA__")
    if not isinstance(blob, (bytes, bytearray)):
        raise ValueError("the pickled CRF carries no CRFsuite model data")
    return bytes(blob)


class _CompatPickler(pickle._Pickler):  # pure-python pickler: lets us emit foreign GLOBALs
    """Writes `PickledRecord`s back under their ORIGINAL class paths so that the file can be
    loaded by stock GECCO (``gecco.crf.ClusterCRF`` etc.), protocol 4 like
    ``ClusterCRF.save`` (``gecco/crf/__init__.py:380-402``)."""

    def save(self, obj, save_persistent_id=True):
        if isinstance(obj, _RecordFactory):
            module, name = obj.glob
            self.save(module)
            self.save(name)
            self.write(pickle.STACK_GLOBAL)
            self.memoize(obj)
            return
        return super().save(obj, save_persistent_id)


def dump_model_dir(record: PickledRecord, directory: Union[str, "pathlib.Path"]) -> None:
    directory = pathlib.Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    buf = io.BytesIO()
    _CompatPickler(buf, protocol=4).dump(record)
    data = buf.getvalue()
    (directory / MODEL_FILENAME).write_bytes(data)
    (directory / f"{MODEL_FILENAME}.md5").write_text(hashlib.md5(data).hexdigest())
