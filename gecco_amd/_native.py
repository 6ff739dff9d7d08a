"""ctypes binding of the C ABI in ``include/gecco_crf.h`` (``gecco_amd/lib/libgecco_crf.so``).

This is the stub a GECCO maintainer would add on the reference side: it
replaces the per-window python-crfsuite crossing at ``gecco/crf/__init__.py:253`` with one
call per batch of contigs.  There is no CPU fallback: if the library cannot be loaded the
import error propagates, and compute calls on a box without a HIP device raise.
"""
import ctypes
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgecco_crf.so")

OK, EINVAL, EFORMAT, ENOMEM, EHIP, ENODEV, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i8p = ctypes.POINTER(ctypes.c_int8)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/gecco_crf.h declares
SIGNATURES = {
    "gecco_crf_last_error": (ctypes.c_char_p, []),
    "gecco_crf_version": (ctypes.c_int, []),
    "gecco_crf_model_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "gecco_crf_model_from_tables": (ctypes.c_int, [_c_f64p, _c_f64p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "gecco_crf_model_free": (None, [_vp]),
    "gecco_crf_model_num_labels": (ctypes.c_int32, [_vp]),
    "gecco_crf_model_num_attrs": (ctypes.c_int32, [_vp]),
    "gecco_crf_model_num_features": (ctypes.c_int32, [_vp]),
    "gecco_crf_model_label_name": (ctypes.c_char_p, [_vp, ctypes.c_int32]),
    "gecco_crf_model_attr_name": (ctypes.c_char_p, [_vp, ctypes.c_int32]),
    "gecco_crf_model_label_id": (ctypes.c_int32, [_vp, ctypes.c_char_p]),
    "gecco_crf_model_attr_id": (ctypes.c_int32, [_vp, ctypes.c_char_p]),
    "gecco_crf_model_map_attrs": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int32, _c_i32p]),
    "gecco_crf_model_state_weights": (ctypes.c_int, [_vp, _c_f64p, _c_u8p]),
    "gecco_crf_model_trans_weights": (ctypes.c_int, [_vp, _c_f64p, _c_u8p]),
    "gecco_crf_device_count": (ctypes.c_int, [_c_i32p]),
    "gecco_crf_windowed_marginals": (
        ctypes.c_int,
        [_vp, ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_int32, _c_f64p],
    ),
    "gecco_crf_marginals_full": (ctypes.c_int, [_vp, ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, _c_f64p, _c_f64p]),
    "gecco_crf_viterbi": (ctypes.c_int, [_vp, ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, _c_i8p, _c_f64p]),
    "gecco_crf_segment": (
        ctypes.c_int,
        [ctypes.c_int32, _c_f64p, _c_u8p, _c_i32p, ctypes.c_int32, ctypes.c_double, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_int32, ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p],
    ),
    "gecco_crf_segment_ex": (
        ctypes.c_int, [ctypes.c_int32, _c_f64p, _c_u8p, _c_i32p, ctypes.c_int32, _vp, _c_i32p, ctypes.c_int32, _c_i32p]
    ),
    "gecco_crf_domain_composition": (
        ctypes.c_int,
        [ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p, ctypes.c_int32, _c_i32p, _c_f64p, ctypes.c_int32, ctypes.c_int32, _c_f64p],
    ),
    "gecco_crf_plan_create": (
        ctypes.c_int, [_vp, ctypes.c_int32, _c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_vp)]
    ),
    "gecco_crf_plan_free": (None, [_vp]),
    "gecco_crf_plan_num_genes": (ctypes.c_int32, [_vp]),
    "gecco_crf_plan_num_windows": (ctypes.c_int64, [_vp]),
    "gecco_crf_plan_num_tiles": (ctypes.c_int32, [_vp]),
    "gecco_crf_plan_tile_out": (ctypes.c_int32, [_vp]),
    "gecco_crf_plan_kernel_name": (ctypes.c_char_p, [_vp]),
    "gecco_crf_plan_run_windowed": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp]),
    "gecco_crf_plan_run_decode": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    "gecco_crf_plan_run_decode_pipelined": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    "gecco_crf_plan_run_marginals_full": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gecco_crf_plan_run_viterbi": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "gecco_crf_plan_run_segment": (
        ctypes.c_int,
        [_vp, _vp, _vp, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp],
    ),
    "gecco_crf_plan_run_segment_ex": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp]),
    "gecco_crf_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "gecco_crf_host_free": (None, [_vp]),
    "gecco_crf_session_create": (ctypes.c_int, [_vp, _c_i32p, ctypes.c_int32, ctypes.POINTER(_vp)]),
    "gecco_crf_session_free": (None, [_vp]),
    "gecco_crf_session_set_chunk_genes": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "gecco_crf_session_set_direct_genes": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "gecco_crf_session_stats": (
        ctypes.c_int,
        [_vp, _c_i32p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _c_f64p, _c_f64p],
    ),
    "gecco_crf_session_stats_ex": (ctypes.c_int, [_vp, _vp]),
    "gecco_crf_session_set_reference_bits": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "gecco_crf_exp_correctly_rounded": (ctypes.c_int, [_c_f64p, ctypes.c_int64, _c_f64p]),
    "gecco_crf_session_windowed": (
        ctypes.c_int,
        # (array arguments as addresses: `ndarray.ctypes.data_as` costs 2 us per array, a quarter of a warm call on one contig)
        [_vp, _vp, ctypes.c_int32, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp],
    ),
    "gecco_crf_session_windowed_degrees": (
        ctypes.c_int,
        [_vp, _vp, ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp],
    ),
    "gecco_crf_session_decode": (
        ctypes.c_int,
        [_vp, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _c_f64p,
         _c_i8p],
    ),
    "gecco_crf_session_clusters": (
        ctypes.c_int,
        [_vp, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, _c_u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _c_f64p, _c_i32p, ctypes.c_int32, _c_i32p, _c_f64p,
         ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)],
    ),
    "gecco_crf_session_clusters_ex": (
        ctypes.c_int,
        [_vp, _c_i32p, ctypes.c_int32, _c_i32p, _c_i32p, _c_u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         _vp, _c_f64p, _c_i32p, ctypes.c_int32, _c_i32p, _c_f64p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)],
    ),
    "gecco_crf_session_clusters_degrees": (
        ctypes.c_int,
        [_vp, _c_i32p, ctypes.c_int32, _c_i32p, _c_u8p, _c_i32p, _c_u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         _vp, _c_f64p, _c_i32p, ctypes.c_int32, _c_i32p, _c_f64p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)],
    ),
    "gecco_crf_session_decode_wire": (
        ctypes.c_int,
        [_vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    ),
    "gecco_crf_session_clusters_wire": (
        ctypes.c_int,
        # (array arguments as plain addresses: a typed ctypes pointer costs 3.5 us to make, an address half of that)
        [_vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, ctypes.c_int64, _vp],
    ),
    "gecco_crf_pack_columns": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp)]),
    "gecco_crf_packed_free": (None, [_vp]),
    "gecco_crf_packed_info": (
        ctypes.c_int,
        [_vp, _c_i32p, _c_i32p, ctypes.POINTER(ctypes.c_int64), _c_i32p, _c_i32p, _c_i32p],
    ),
    "gecco_crf_packed_contig_ptr": (_vp, [_vp]),
    "gecco_crf_packed_gene_ptr": (_vp, [_vp]),
    "gecco_crf_packed_attr_id": (_vp, [_vp]),
    "gecco_crf_packed_annotated": (_vp, [_vp]),
    "gecco_crf_packed_gene_row": (_vp, [_vp]),
    "gecco_crf_packed_row_gene": (_vp, [_vp]),
    "gecco_crf_packed_row_order": (_vp, [_vp]),
    "gecco_crf_packed_row_ptr": (_vp, [_vp]),
    "gecco_crf_packed_marker_ptr": (_vp, [_vp]),
    "gecco_crf_packed_marker_id": (_vp, [_vp]),
    "gecco_crf_cluster_rows_build": (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_i32p, ctypes.c_int32, _c_f64p, _vp, ctypes.POINTER(_vp)]),
    "gecco_crf_cluster_rows_free": (None, [_vp]),
    "gecco_crf_cluster_rows_start": (_vp, [_vp]),
    "gecco_crf_cluster_rows_end": (_vp, [_vp]),
    "gecco_crf_cluster_rows_average_p": (_vp, [_vp]),
    "gecco_crf_cluster_rows_max_p": (_vp, [_vp]),
    "gecco_crf_cluster_rows_strings": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "gecco_crf_exact_mean": (ctypes.c_double, [_c_f64p, ctypes.c_int64]),
    "gecco_crf_gather_f64": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, _vp]),
    "gecco_crf_packed_order_info": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "gecco_crf_tsv_format": (
        ctypes.c_int,
        [ctypes.c_int64, ctypes.c_int32, _c_i32p, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_char_p, ctypes.POINTER(_vp),
         ctypes.POINTER(ctypes.c_int64)],
    ),
    "gecco_crf_buffer_free": (None, [_vp]),
    "gecco_crf_plan_time_windowed": (
        ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]
    ),
    "gecco_crf_plan_time_decode_pipelined": (
        ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]
    ),
    "gecco_crf_plan_viterbi_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]),
}

_lib = None


def _elf_dynamic_strings(path: str, tags=(1, 14)) -> dict:
    """DT_NEEDED (1) / DT_SONAME (14) strings of a 64-bit little-endian ELF shared object, read from the file (nothing is loaded)."""
    import struct

    out = {t: [] for t in tags}
    with open(path, "rb") as fh:
        hdr = fh.read(64)
        if hdr[:4] != b"\x7fELF" or hdr[4] != 2 or hdr[5] != 1:
            return out
        shoff, = struct.unpack_from("<Q", hdr, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", hdr, 0x3A)
        secs = []
        for i in range(shnum):
            fh.seek(shoff + i * shentsize)
            sh = fh.read(shentsize)
            _, typ, _, _, off, size, link = struct.unpack_from("<IIQQQQI", sh, 0)
            secs.append((typ, off, size, link))
        for typ, off, size, link in secs:
            if typ != 6:  # SHT_DYNAMIC
                continue
            _, stroff, strsize, _ = secs[link]
            fh.seek(stroff)
            strtab = fh.read(strsize)
            fh.seek(off)
            dyn = fh.read(size)
            for k in range(0, len(dyn) - 15, 16):
                tag, val = struct.unpack_from("<qQ", dyn, k)
                if tag == 0:
                    break
                if tag in out:
                    out[tag].append(strtab[val:strtab.index(b"\0", val)].decode())
    return out


def _preload_hip_runtime(lib_path: str) -> Optional[str]:
    """A PyTorch wheel carries its own copy of libamdhip64, and whichever copy a process loads first serves everybody: loaded
    second, torch's copy finds no device.  `GECCO_AMD_HIP_RUNTIME` decides what a process that has NOT imported torch yet does:

    * unset / ``auto``: the wheel's copy is loaded ahead of libgecco_crf.so ONLY when its SONAME equals the libamdhip64 SONAME
      libgecco_crf.so was linked against (same ABI major: one runtime in the process, whatever the import order); a different
      SONAME gets a warning and the system's runtime (a later `import torch` in that process will then not see the device);
    * ``system``: never preload;  * ``torch``: preload whatever the wheel carries (the caller vouches for it).

    A process that has imported torch already has made the choice.  Returns the path that was loaded, or None."""
    import sys
    import warnings

    mode = os.environ.get("GECCO_AMD_HIP_RUNTIME", "auto").lower()
    if "torch" in sys.modules or mode == "system":
        return None
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    if mode != "torch":
        try:
            wheel = _elf_dynamic_strings(cand)[14]
            needed = [n for n in _elf_dynamic_strings(lib_path)[1] if n.startswith("libamdhip64")]
        except (OSError, ValueError, IndexError) as err:
            warnings.warn(f"gecco_amd: could not compare the HIP runtime of the installed torch wheel with the one {lib_path} was linked "
                          f"against ({err}); keeping the system's runtime", RuntimeWarning, stacklevel=3)
            return None
        if not wheel or not needed or wheel[0] != needed[0]:
            warnings.warn(f"gecco_amd: the installed torch wheel carries {wheel[0] if wheel else 'an unnamed libamdhip64'}, "
                          f"libgecco_crf.so is linked against {needed[0] if needed else 'no libamdhip64'}: keeping the system's HIP "
                          "runtime (import torch BEFORE gecco_amd if this process needs both, or set GECCO_AMD_HIP_RUNTIME=torch)",
                          RuntimeWarning, stacklevel=3)
            return None
    try:
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError as err:
        warnings.warn(f"gecco_amd: could not load {cand} ({err}); the system's HIP runtime serves", RuntimeWarning, stacklevel=3)
        return None
    return cand


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the native library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GECCO_CRF_LIBRARY") or LIB_PATH  # the variable is for A/B runs of two builds
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build it with `python -m gecco_amd.build` (hipcc, gfx950). "
            "gecco_amd has no CPU fallback."
        )
    _preload_hip_runtime(p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{msg} (gecco_crf status {code})")
        self.code = code


def _check(rc: int) -> None:
    if rc == OK:
        return
    msg = load_library().gecco_crf_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EFORMAT:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise NativeError(rc, msg)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def device_count() -> int:
    n = ctypes.c_int32(0)
    _check(load_library().gecco_crf_device_count(ctypes.byref(n)))
    return n.value


class Model:
    """Handle on a parsed CRF model (immutable, thread-safe)."""

    def __init__(self, handle):
        self._h = handle
        self._lib = load_library()

    @classmethod
    def from_lcrf(cls, blob: bytes) -> "Model":
        lib = load_library()
        h = _vp()
        _check(lib.gecco_crf_model_load(blob, len(blob), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_tables(cls, state: np.ndarray, trans: np.ndarray) -> "Model":
        lib = load_library()
        state = np.ascontiguousarray(state, dtype=np.float64)
        trans = np.ascontiguousarray(trans, dtype=np.float64)
        A, L = state.shape
        assert trans.shape == (L, L)
        h = _vp()
        _check(lib.gecco_crf_model_from_tables(_ptr(state, _c_f64p), _ptr(trans, _c_f64p), A, L, ctypes.byref(h)))
        return cls(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gecco_crf_model_free(h)

    @property
    def num_labels(self) -> int:
        return self._lib.gecco_crf_model_num_labels(self._h)

    @property
    def num_attrs(self) -> int:
        return self._lib.gecco_crf_model_num_attrs(self._h)

    @property
    def num_features(self) -> int:
        return self._lib.gecco_crf_model_num_features(self._h)

    def labels(self):
        return [self._lib.gecco_crf_model_label_name(self._h, i).decode() for i in range(self.num_labels)]

    def attrs(self):
        return [self._lib.gecco_crf_model_attr_name(self._h, i).decode() for i in range(self.num_attrs)]

    def label_id(self, name: str) -> int:
        return self._lib.gecco_crf_model_label_id(self._h, name.encode())

    def attr_id(self, name: str) -> int:
        return self._lib.gecco_crf_model_attr_id(self._h, name.encode())

    def map_attrs(self, names: Sequence[str]) -> np.ndarray:
        n = len(names)
        arr = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
        ids = np.empty(max(n, 1), dtype=np.int32)
        _check(self._lib.gecco_crf_model_map_attrs(self._h, arr, n, _ptr(ids, _c_i32p)))
        return ids[:n]

    def state_weights(self):
        A, L = self.num_attrs, self.num_labels
        w = np.zeros((A, L), dtype=np.float64)
        present = np.zeros((A, L), dtype=np.uint8)
        _check(self._lib.gecco_crf_model_state_weights(self._h, _ptr(w, _c_f64p), _ptr(present, _c_u8p)))
        return w, present.astype(bool)

    def trans_weights(self):
        L = self.num_labels
        w = np.zeros((L, L), dtype=np.float64)
        present = np.zeros((L, L), dtype=np.uint8)
        _check(self._lib.gecco_crf_model_trans_weights(self._h, _ptr(w, _c_f64p), _ptr(present, _c_u8p)))
        return w, present.astype(bool)

    # ---- one-shot compute (host buffers) ----
    def windowed_marginals(self, contig_ptr, gene_ptr, attr_id, window, step=1, label=1, pad=True, device=0):
        contig_ptr, gene_ptr, attr_id = _i32(contig_ptr), _i32(gene_ptr), _i32(attr_id)
        n = int(contig_ptr[-1]) if len(contig_ptr) else 0
        out = np.zeros(max(n, 1), dtype=np.float64)
        if attr_id.size == 0:
            attr_id = np.zeros(1, dtype=np.int32)
        _check(
            self._lib.gecco_crf_windowed_marginals(
                self._h, device, _ptr(contig_ptr, _c_i32p), max(len(contig_ptr) - 1, 0), _ptr(gene_ptr, _c_i32p),
                _ptr(attr_id, _c_i32p), int(window), int(step), int(label), int(bool(pad)), _ptr(out, _c_f64p),
            )
        )
        return out[:n]

    def marginals_full(self, contig_ptr, gene_ptr, attr_id, device=0):
        contig_ptr, gene_ptr, attr_id = _i32(contig_ptr), _i32(gene_ptr), _i32(attr_id)
        n, nc, L = int(contig_ptr[-1]), len(contig_ptr) - 1, self.num_labels
        marg = np.zeros((max(n, 1), L), dtype=np.float64)
        ln = np.zeros(max(nc, 1), dtype=np.float64)
        if attr_id.size == 0:
            attr_id = np.zeros(1, dtype=np.int32)
        _check(
            self._lib.gecco_crf_marginals_full(
                self._h, device, _ptr(contig_ptr, _c_i32p), nc, _ptr(gene_ptr, _c_i32p), _ptr(attr_id, _c_i32p),
                _ptr(marg, _c_f64p), _ptr(ln, _c_f64p),
            )
        )
        return marg[:n], ln[:nc]

    def viterbi(self, contig_ptr, gene_ptr, attr_id, device=0, want_score=True):
        """Best label path per contig; with `want_score=False` returns (labels, None) and 2-label
        models take the cheaper score-difference form of the recursion."""
        contig_ptr, gene_ptr, attr_id = _i32(contig_ptr), _i32(gene_ptr), _i32(attr_id)
        n, nc = int(contig_ptr[-1]), len(contig_ptr) - 1
        y = np.zeros(max(n, 1), dtype=np.int8)
        sc = np.zeros(max(nc, 1), dtype=np.float64) if want_score else None
        if attr_id.size == 0:
            attr_id = np.zeros(1, dtype=np.int32)
        _check(
            self._lib.gecco_crf_viterbi(
                self._h, device, _ptr(contig_ptr, _c_i32p), nc, _ptr(gene_ptr, _c_i32p), _ptr(attr_id, _c_i32p),
                _ptr(y, _c_i8p), _ptr(sc, _c_f64p) if want_score else None,
            )
        )
        return y[:n], (sc[:nc] if want_score else None)


def domain_composition(seg, dom_ptr, dom_col, dom_weight, n_cols, normalize=True, device=0) -> np.ndarray:
    """Dense (n_seg, n_cols) weighted domain compositions of the clusters in `seg` (gecco_crf_domain_composition)."""
    lib = load_library()
    seg = np.ascontiguousarray(seg, dtype=np.int32).reshape(-1, 4)
    dom_ptr, dom_col = _i32(dom_ptr), _i32(dom_col)
    dom_weight = np.ascontiguousarray(dom_weight, dtype=np.float64)
    if dom_col.size == 0:
        dom_col, dom_weight = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.float64)
    out = np.zeros((len(seg), int(n_cols)), dtype=np.float64)
    seg_buf = seg if len(seg) else np.zeros((1, 4), dtype=np.int32)
    out_buf = out if out.size else np.zeros(1, dtype=np.float64)
    _check(lib.gecco_crf_domain_composition(
        device, _ptr(seg_buf, _c_i32p), len(seg), _ptr(dom_ptr, _c_i32p), len(dom_ptr) - 1, _ptr(dom_col, _c_i32p),
        _ptr(dom_weight, _c_f64p), int(n_cols), int(bool(normalize)), _ptr(out_buf, _c_f64p)))
    return out


class RefineParams(ctypes.Structure):
    """``gecco_crf_refine_params``: ClusterRefiner's parameters (gecco/refine.py:75-116)."""
    _fields_ = [("threshold", ctypes.c_double), ("average_threshold", ctypes.c_double), ("criterion", ctypes.c_int32),
                ("n_cds", ctypes.c_int32), ("n_biopfams", ctypes.c_int32), ("edge_distance", ctypes.c_int32),
                ("trim", ctypes.c_int32), ("carry_state", ctypes.c_int32), ("marker_ptr", _vp), ("marker_id", _vp)]


CRITERIA = {"gecco": 0, "antismash": 1}


def refine_params(criterion="gecco", threshold=0.8, n_cds=5, n_biopfams=5, average_threshold=0.6, edge_distance=0, trim=True,
                  carry_state=False, marker_ptr=None, marker_id=None, keep=None) -> RefineParams:
    """The C struct; `marker_ptr` / `marker_id` are host int32 arrays (parked in `keep`) or integer device addresses."""
    if criterion not in CRITERIA:
        raise ValueError(f"Unknown cluster filtering criterion: {criterion}")  # refine.py:165
    q = RefineParams(float(threshold), float(average_threshold), CRITERIA[criterion], int(n_cds), int(n_biopfams),
                     int(edge_distance), int(bool(trim)), int(bool(carry_state)), None, None)
    for name, arr in (("marker_ptr", marker_ptr), ("marker_id", marker_id)):
        if arr is None:
            continue
        if isinstance(arr, int):
            setattr(q, name, arr or None)
            continue
        a = _i32(arr)
        if a.size == 0:
            a = np.zeros(1, dtype=np.int32)
        if keep is not None:
            keep.append(a)
        setattr(q, name, a.ctypes.data)
    return q


def segment(p, annotated, contig_ptr, threshold=0.8, n_cds=3, edge_distance=0, trim=True, device=0,
            carry_state=False, criterion="gecco", n_biopfams=5, average_threshold=0.6, marker_ptr=None,
            marker_id=None) -> np.ndarray:
    """Cluster rows (contig, number, first gene, last gene + 1) of per-gene probabilities.  `carry_state`:
    False = one grouper per contig (the CLI's ``iter_clusters`` call per contig), True = one grouper
    over all contigs (a single ``iter_clusters`` call).  `criterion="antismash"` needs the genes' marker domains
    (`marker_ptr[n_genes+1]`, `marker_id`: indices into the caller's marker list, refine.py:157-163)."""
    lib = load_library()
    p = np.ascontiguousarray(p, dtype=np.float64)
    annotated = np.ascontiguousarray(annotated, dtype=np.uint8)
    contig_ptr = _i32(contig_ptr)
    cap = max(1, len(p))
    seg = np.zeros((cap, 4), dtype=np.int32)
    n_seg = ctypes.c_int32(0)
    keep = []
    q = refine_params(criterion, threshold, n_cds, n_biopfams, average_threshold, edge_distance, trim, carry_state, marker_ptr,
                      marker_id, keep)
    _check(lib.gecco_crf_segment_ex(device, _ptr(p, _c_f64p), _ptr(annotated, _c_u8p), _ptr(contig_ptr, _c_i32p),
                                    len(contig_ptr) - 1, ctypes.byref(q), _ptr(seg, _c_i32p), cap, ctypes.byref(n_seg)))
    return seg[: n_seg.value].copy()


# ---- columnar host side (gecco_crf_pack_columns / gecco_crf_cluster_rows_build) ----------------------
class _Strings(ctypes.Structure):
    _fields_ = [("data", _vp), ("offsets", _vp)]


class _TableColumns(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_int64), ("sequence_id", _Strings), ("protein_id", _Strings), ("domain", _Strings),
                ("start", _vp), ("domain_start", _vp), ("n_genes", ctypes.c_int64), ("gene_sequence_id", _Strings),
                ("gene_protein_id", _Strings), ("gene_start", _vp), ("n_markers", ctypes.c_int64), ("markers", _Strings)]


def _view(ptr, n, dtype, owner):
    """numpy view over `n` items of library-owned memory; `owner` (the handle wrapper) is kept alive by it."""
    n = int(n)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype, count=n)


def _strings_arg(col, keep):
    """(data uint8, offsets int64) of a string column -> the C struct; arrays are parked in `keep`."""
    data = np.ascontiguousarray(col.data, dtype=np.uint8)
    off = np.ascontiguousarray(col.offsets, dtype=np.int64)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    keep.extend((data, off))
    return _Strings(data.ctypes.data, off.ctypes.data)


def _i64_arg(a, keep):
    a = np.ascontiguousarray(a, dtype=np.int64)
    if a.size == 0:
        a = np.zeros(1, dtype=np.int64)
    keep.append(a)
    return a.ctypes.data


class PackedTables:
    """CSR batch + row bookkeeping of a feature table (and gene table), built by `gecco_crf_pack_columns`.
    The arrays are views of library-owned memory (pinned when a device is present)."""

    def __init__(self, model: "Model", f_sequence_id, f_protein_id, f_start, f_domain, f_domain_start,
                 g_sequence_id=None, g_protein_id=None, g_start=None, markers=None):
        self._lib = load_library()
        keep = []
        t = _TableColumns()
        t.n_rows = len(f_protein_id)
        t.sequence_id = _strings_arg(f_sequence_id, keep)
        t.protein_id = _strings_arg(f_protein_id, keep)
        t.domain = _strings_arg(f_domain, keep)
        t.start = _i64_arg(f_start, keep)
        t.domain_start = _i64_arg(f_domain_start, keep)
        t.n_genes = 0 if g_protein_id is None else len(g_protein_id)
        if g_protein_id is not None:
            t.gene_sequence_id = _strings_arg(g_sequence_id, keep)
            t.gene_protein_id = _strings_arg(g_protein_id, keep)
            t.gene_start = _i64_arg(g_start, keep)
        t.n_markers = 0 if markers is None else len(markers)
        if t.n_markers:
            t.markers = _strings_arg(markers, keep)
        self._cols, self._keep = t, keep  # the cluster-row builder reads the same columns again
        h = _vp()
        _check(self._lib.gecco_crf_pack_columns(model._h, ctypes.byref(t), ctypes.byref(h)))
        self._h = h
        ng, nc, dup, unl, pin = (ctypes.c_int32(0) for _ in range(5))
        nnz = ctypes.c_int64(0)
        _check(self._lib.gecco_crf_packed_info(h, ctypes.byref(ng), ctypes.byref(nc), ctypes.byref(nnz), ctypes.byref(dup),
                                               ctypes.byref(unl), ctypes.byref(pin)))
        self.n_genes, self.n_contigs, self.nnz = ng.value, nc.value, nnz.value
        self.n_rows = int(t.n_rows)
        self.n_duplicate_gene_ids, self.n_unlisted_proteins, self.pinned = dup.value, unl.value, bool(pin.value)
        L = self._lib
        self.contig_ptr = _view(L.gecco_crf_packed_contig_ptr(h), self.n_contigs + 1, np.int32, self)
        self.gene_ptr = _view(L.gecco_crf_packed_gene_ptr(h), self.n_genes + 1, np.int32, self)
        self.attr_id = _view(L.gecco_crf_packed_attr_id(h), self.nnz, np.int32, self)
        self.annotated = _view(L.gecco_crf_packed_annotated(h), self.n_genes, np.uint8, self)
        self.gene_row = _view(L.gecco_crf_packed_gene_row(h), self.n_genes, np.int64, self)
        self.row_gene = _view(L.gecco_crf_packed_row_gene(h), self.n_rows, np.int32, self)
        self.row_order = _view(L.gecco_crf_packed_row_order(h), self.n_rows, np.int64, self)
        self.row_ptr = _view(L.gecco_crf_packed_row_ptr(h), self.n_genes + 1, np.int64, self)
        self.marker_ptr = self.marker_id = None  # per gene its marker domains (antismash criterion), when asked for
        if t.n_markers:
            self.marker_ptr = _view(L.gecco_crf_packed_marker_ptr(h), self.n_genes + 1, np.int32, self)
            if self.n_genes == 0:
                self.marker_ptr = np.zeros(1, dtype=np.int32)
            self.marker_id = _view(L.gecco_crf_packed_marker_id(h), int(self.marker_ptr[-1]), np.int32, self)
        if self.n_contigs == 0:
            self.contig_ptr = np.zeros(1, dtype=np.int32)
        if self.n_genes == 0:
            self.gene_ptr = np.zeros(1, dtype=np.int32)
            self.row_ptr = np.zeros(1, dtype=np.int64)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gecco_crf_packed_free(h)

    def order_info(self, gene_start: np.ndarray, gene_end: np.ndarray):
        """(rows_in_order, refiner_order_differs) from the gene table's `start` / `end` columns (`gecco_crf_packed_order_info`):
        whether the genes in scoring order are the gene table's rows 0 .. n - 1, and whether the refiner's (start, end) order
        differs from the scoring order somewhere (equal starts with decreasing ends)."""
        gs = np.ascontiguousarray(gene_start, dtype=np.int64)
        ge = np.ascontiguousarray(gene_end, dtype=np.int64)
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        _check(self._lib.gecco_crf_packed_order_info(self._h, gs.ctypes.data, ge.ctypes.data, min(gs.size, ge.size), ctypes.byref(a),
                                                     ctypes.byref(b)))
        return bool(a.value), bool(b.value)

    def cluster_rows(self, seg, seg_p, seg_off, gene_end, feature_end) -> dict:
        """Columns of clusters.tsv for `seg` rows: dict of numpy arrays and (data, offsets) string columns."""
        seg = np.ascontiguousarray(seg, dtype=np.int32).reshape(-1, 4)
        seg_p = np.ascontiguousarray(seg_p, dtype=np.float64)
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        keep = []
        h = _vp()
        k = len(seg)
        segb = seg if k else np.zeros((1, 4), dtype=np.int32)
        spb = seg_p if seg_p.size else np.zeros(1, dtype=np.float64)
        _check(self._lib.gecco_crf_cluster_rows_build(
            self._h, ctypes.byref(self._cols), _i64_arg(gene_end if gene_end is not None else [], keep),
            _i64_arg(feature_end if feature_end is not None else [], keep), _ptr(segb, _c_i32p), k, _ptr(spb, _c_f64p),
            seg_off.ctypes.data, ctypes.byref(h)))
        try:
            L = self._lib
            out = {
                "start": _view(L.gecco_crf_cluster_rows_start(h), k, np.int64, None).copy(),
                "end": _view(L.gecco_crf_cluster_rows_end(h), k, np.int64, None).copy(),
                "average_p": _view(L.gecco_crf_cluster_rows_average_p(h), k, np.float64, None).copy(),
                "max_p": _view(L.gecco_crf_cluster_rows_max_p(h), k, np.float64, None).copy(),
            }
            for which, name in enumerate(("sequence_id", "cluster_id", "proteins", "domains")):
                d, o = _vp(), _vp()
                _check(L.gecco_crf_cluster_rows_strings(h, which, ctypes.byref(d), ctypes.byref(o)))
                off = _view(o.value, k + 1, np.int64, None).copy()
                out[name] = (_view(d.value, int(off[-1]) if k else 0, np.uint8, None).copy(), off)
        finally:
            self._lib.gecco_crf_cluster_rows_free(h)
        return out


def tsv_format(header: str, columns) -> bytes:
    """TSV text of `columns` (each a (data uint8, offsets int64) string column, an int64 array or a float64 array)
    below `header`: floats as Python's repr() writes them, NaN as an empty field (`gecco_crf_tsv_format`)."""
    lib = load_library()
    keep, kinds, data, offs = [], [], [], []
    n_rows = None
    for col in columns:
        if isinstance(col, tuple):
            d = np.ascontiguousarray(col[0], dtype=np.uint8)
            o = np.ascontiguousarray(col[1], dtype=np.int64)
            if d.size == 0:
                d = np.zeros(1, dtype=np.uint8)
            keep.extend((d, o))
            kinds.append(0)
            data.append(d.ctypes.data)
            offs.append(o.ctypes.data)
            n = len(o) - 1
        else:
            a = np.asarray(col)
            a = np.ascontiguousarray(a, dtype=np.float64 if a.dtype.kind == "f" else np.int64)
            if a.size == 0:
                a = np.zeros(1, dtype=a.dtype)
            keep.append(a)
            kinds.append(2 if a.dtype.kind == "f" else 1)
            data.append(a.ctypes.data)
            offs.append(None)
            n = len(np.asarray(col))
        if n_rows is None:
            n_rows = n
        elif n != n_rows:
            raise ValueError("columns of different lengths")
    nc = len(kinds)
    k = np.array(kinds or [0], dtype=np.int32)
    dp = (_vp * max(nc, 1))(*data)
    op = (_vp * max(nc, 1))(*offs)
    out, out_len = _vp(), ctypes.c_int64(0)
    _check(lib.gecco_crf_tsv_format(n_rows or 0, nc, _ptr(k, _c_i32p), dp, op, header.encode("utf-8"), ctypes.byref(out),
                                    ctypes.byref(out_len)))
    try:
        return ctypes.string_at(out.value, out_len.value)
    finally:
        lib.gecco_crf_buffer_free(out)


def gather_f64(src: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """src[idx] on several host threads (`gecco_crf_gather_f64`; idx int32, src float64)."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(idx.size, dtype=np.float64)
    if idx.size:
        _check(load_library().gecco_crf_gather_f64(src.ctypes.data, src.size, idx.ctypes.data, idx.size, out.ctypes.data))
    return out


def exact_mean(values) -> float:
    """``statistics.mean`` of the non-NaN values (exact sum, one rounding), natively."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    if v.size == 0:
        return float("nan")
    return float(load_library().gecco_crf_exact_mean(_ptr(v, _c_f64p), v.size))


def exp_correctly_rounded(x) -> np.ndarray:
    """The double nearest to exp(x), elementwise (`gecco_crf_exp_correctly_rounded`: what reference-bits mode puts in libm's place)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if x.size:
        _check(load_library().gecco_crf_exp_correctly_rounded(_ptr(x.reshape(-1), _c_f64p), x.size, _ptr(out.reshape(-1), _c_f64p)))
    return out


class _PinnedBlock:
    """Owner of one gecco_crf_host_alloc block (freed when the last array over it dies)."""

    def __init__(self, nbytes: int):
        self._lib = load_library()
        ptr = _vp()
        _check(self._lib.gecco_crf_host_alloc(max(int(nbytes), 1), ctypes.byref(ptr)))
        self.ptr = ptr.value
        self.nbytes = max(int(nbytes), 1)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            self._lib.gecco_crf_host_free(ptr)


def pinned_empty(shape, dtype) -> np.ndarray:
    """numpy array over pinned (page-locked, device-visible) host memory: the batch driver copies to
    and from such arrays asynchronously, so uploads, kernels and downloads of a batch overlap."""
    dtype = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    block = _PinnedBlock(n * dtype.itemsize)
    buf = (ctypes.c_char * block.nbytes).from_address(block.ptr)
    buf._pinned_block = block  # numpy keeps `buf` (the buffer exporter) alive, `buf` keeps the block
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_copy(a, dtype=None) -> np.ndarray:
    a = np.asarray(a, dtype=dtype)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class DecodeStream:
    """The pipelined decode (`gecco_crf_plan_run_decode_pipelined`) over a sequence of resident batches on one HIP stream:
    `submit` enqueues the window marginals of a batch and the Viterbi labels of the batch submitted before it (one launch
    when both qualify), `flush` the labels of the last one.  Two of these on two streams, batches alternating between
    them, keep two launches in flight (bench.py's schedule).  The caller keeps a batch's device arrays alive until the
    next `submit` / `flush` has been enqueued."""

    def __init__(self, stream: int = 0):
        self.stream = stream
        self._prev = None  # (plan, address of its label buffer)

    def submit(self, plan: "Plan", d_gene_ptr: int, d_attr_id: int, d_p_out: int, d_y_out: int, label: int = 1) -> None:
        prev_plan, prev_y = self._prev if self._prev is not None else (None, 0)
        plan.run_decode_pipelined(d_gene_ptr, d_attr_id, d_p_out, prev_plan, prev_y, label, self.stream)
        self._prev = (plan, d_y_out)

    def flush(self) -> None:
        if self._prev is not None:
            self._prev[0].flush_decode_pipelined(self._prev[1], self.stream)
            self._prev = None


def degree_bytes(gene_ptr) -> np.ndarray:
    """The wire format of `Session.windowed_marginals(degree=...)`: domain counts of the genes as bytes."""
    d = np.diff(np.asarray(gene_ptr, dtype=np.int64))
    if d.size and (d.min() < 0 or d.max() > 255):
        raise ValueError("degree bytes need 0 <= domains per gene <= 255")
    return np.ascontiguousarray(d, dtype=np.uint8) if d.size else np.zeros(1, dtype=np.uint8)


class SessionStatsStruct(ctypes.Structure):
    """``gecco_crf_session_stats_t``"""
    _fields_ = [("n_chunks", ctypes.c_int32), ("n_devices", ctypes.c_int32), ("direct", ctypes.c_int32), ("host_threads", ctypes.c_int32),
                ("h2d_bytes", ctypes.c_int64), ("d2h_bytes", ctypes.c_int64), ("host_plan_seconds", ctypes.c_double),
                ("host_issue_seconds", ctypes.c_double), ("wall_seconds", ctypes.c_double)]


class Session:
    """Batch driver over one or several devices (include/gecco_crf.h, `gecco_crf_session_*`): host
    arrays in, host arrays out; contig chunks are dealt to the devices longest-first and pipelined
    (upload / compute / download) on each of them."""

    def __init__(self, model: "Model", devices: Sequence[int] = (0,)):
        self._lib = load_library()
        self.model = model  # the session must not outlive its model
        devs = _i32(list(devices))
        h = _vp()
        _check(self._lib.gecco_crf_session_create(model._h, _ptr(devs, _c_i32p), len(devs), ctypes.byref(h)))
        self._h = h
        self.devices = [int(d) for d in devs]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gecco_crf_session_free(h)

    def set_chunk_genes(self, genes: int) -> None:
        _check(self._lib.gecco_crf_session_set_chunk_genes(self._h, int(genes)))

    def set_reference_bits(self, on: bool = True) -> None:
        """Windowed marginals in CRFsuite's own operation order with a correctly rounded exp: the reference's output files bit
        for bit, at about six times the fast kernels' time (`gecco_crf_session_set_reference_bits`)."""
        _check(self._lib.gecco_crf_session_set_reference_bits(self._h, int(bool(on))))

    def set_direct_genes(self, genes: int) -> None:
        """Largest batch (genes) that takes the direct path -- one chunk, kernels on pinned host memory, no copy commands
        (`gecco_crf_session_set_direct_genes`; 0: never; -1: the defaults -- every one-chunk batch on a one-device session,
        131 072 genes on a session over several devices, 65 536 for cluster calls)."""
        _check(self._lib.gecco_crf_session_set_direct_genes(self._h, int(genes)))

    def stats(self) -> dict:
        """Figures of the last batch (`gecco_crf_session_stats_ex`)."""
        st = SessionStatsStruct()
        _check(self._lib.gecco_crf_session_stats_ex(self._h, ctypes.addressof(st)))
        return {name: getattr(st, name) for name, _ in SessionStatsStruct._fields_}

    @staticmethod
    def _csr(contig_ptr, gene_ptr, attr_id):
        contig_ptr, gene_ptr, attr_id = _i32(contig_ptr), _i32(gene_ptr), _i32(attr_id)
        if attr_id.size == 0:
            attr_id = np.zeros(1, dtype=np.int32)
        n = int(contig_ptr[-1]) if len(contig_ptr) else 0
        return contig_ptr, gene_ptr, attr_id, n, max(len(contig_ptr) - 1, 0)

    def windowed_marginals(self, contig_ptr, gene_ptr, attr_id, window, step=1, label=1, pad=True, out=None, degree=None):
        """`degree`: the genes' domain counts as uint8 (== numpy.diff(gene_ptr)): they cross PCIe instead of the row pointers
        (gecco_crf_session_windowed_degrees); `degree_bytes(gene_ptr)` makes them."""
        contig_ptr, gene_ptr, attr_id, n, nc = self._csr(contig_ptr, gene_ptr, attr_id)
        if out is None:
            out = np.empty(max(n, 1), dtype=np.float64)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size >= n
        if degree is not None:
            assert degree.dtype == np.uint8 and degree.flags.c_contiguous and degree.size >= n
            _check(self._lib.gecco_crf_session_windowed_degrees(self._h, contig_ptr.ctypes.data, nc, gene_ptr.ctypes.data,
                                                                degree.ctypes.data if degree.size else None, attr_id.ctypes.data,
                                                                int(window), int(step), int(label), int(bool(pad)), out.ctypes.data))
            return out[:n]
        _check(self._lib.gecco_crf_session_windowed(self._h, contig_ptr.ctypes.data, nc, gene_ptr.ctypes.data, attr_id.ctypes.data,
                                                    int(window), int(step), int(label), int(bool(pad)), out.ctypes.data))
        return out[:n]

    def decode(self, contig_ptr, gene_ptr, attr_id, window, step=1, label=1, pad=True, out_p=None, out_y=None, degree=None,
               labels=True):
        """Windowed marginals + whole-contig Viterbi labels of a batch in host memory; `out_p` / `out_y`: caller buffers
        (pinned ones make the downloads asynchronous).  The compact wire format (gecco_crf_session_decode_wire): `degree`
        = the genes' domain counts as uint8 (`degree_bytes(gene_ptr)`), sent instead of the row pointers; `attr_id` as a
        uint16 array (a model with at most 65536 attributes) crosses PCIe as it is.  `labels=False`: marginals only
        (returns (p, None))."""
        attr16 = None
        if isinstance(attr_id, np.ndarray) and attr_id.dtype == np.uint16:
            attr16 = np.ascontiguousarray(attr_id) if attr_id.size else np.zeros(1, dtype=np.uint16)
            attr_id = np.zeros(1, dtype=np.int32)
        contig_ptr, gene_ptr, attr_id, n, nc = self._csr(contig_ptr, gene_ptr, attr_id)
        p = np.empty(max(n, 1), dtype=np.float64) if out_p is None else out_p
        y = (np.empty(max(n, 1), dtype=np.int8) if out_y is None else out_y) if labels else None
        assert p.dtype == np.float64 and p.flags.c_contiguous and p.size >= n
        assert y is None or (y.dtype == np.int8 and y.flags.c_contiguous and y.size >= n)
        if degree is not None:
            assert degree.dtype == np.uint8 and degree.flags.c_contiguous and degree.size >= n
        adr = lambda a: None if a is None or a.size == 0 else a.ctypes.data  # noqa: E731
        _check(self._lib.gecco_crf_session_decode_wire(
            self._h, adr(contig_ptr), nc, adr(gene_ptr), adr(degree), adr(attr_id) if attr16 is None else None, adr(attr16),
            int(window), int(step), int(label), int(bool(pad)), p.ctypes.data, None if y is None else y.ctypes.data))
        return p[:n], (None if y is None else y[:n])

    def clusters(self, contig_ptr, gene_ptr, attr_id, annotated, window, step=1, label=1, pad=True, threshold=0.8, n_cds=3,
                 edge_distance=0, trim=True, want_p=False, want_seg_p=True, p_out=None, criterion="gecco", n_biopfams=5,
                 average_threshold=0.6, marker_ptr=None, marker_id=None, degree=None, seg_p_out=None):
        """Windowed marginals + cluster calls in one pass; the probabilities stay on the device unless
        `want_p` / `p_out`.  Returns (seg rows (k, 4), seg_p, seg_off, p or None): `seg_p[seg_off[i]:seg_off[i+1]]`
        are the probabilities of the genes of row i.  `degree` (uint8, = diff(gene_ptr)): the degree-byte wire format.
        `seg_p_out`: a caller buffer of n doubles for the rows' probabilities (a pinned one is filled by the copy engine at
        full rate; a fresh pageable array costs a page fault per 4 KB); the returned seg_p is then a view of it.
        `attr_id` may be a uint16 array (a model with at most 65536 attributes): it then crosses PCIe as it is."""
        attr16 = None
        if isinstance(attr_id, np.ndarray) and attr_id.dtype == np.uint16:
            attr16 = np.ascontiguousarray(attr_id) if attr_id.size else np.zeros(1, dtype=np.uint16)
            attr_id = np.zeros(1, dtype=np.int32)
        contig_ptr, gene_ptr, attr_id, n, nc = self._csr(contig_ptr, gene_ptr, attr_id)
        if annotated is None:  # (with `degree`: annotated iff the gene has a domain the model knows)
            if degree is None:
                raise ValueError("clusters: `annotated` may only be left out together with `degree`")
        else:
            annotated = np.ascontiguousarray(annotated, dtype=np.uint8)
            if annotated.size == 0:
                annotated = np.zeros(1, dtype=np.uint8)
        if p_out is None and want_p:
            p_out = np.empty(max(n, 1), dtype=np.float64)
        cap = min(n, n // 2 + nc) + 1
        # the row buffers are kept between calls (no 24 MB of fresh pages per batch); a second thread inside this method on the
        # same session takes fresh ones
        held = self.__dict__.pop("_row_buffers", None)
        if held is None or held[0].shape[0] < cap:
            held = (np.empty((cap, 4), dtype=np.int32), np.zeros(cap + 1, dtype=np.int64))
        seg, seg_off = held[0][:cap], held[1][: cap + 1]
        if p_out is not None:
            assert p_out.dtype == np.float64 and p_out.flags.c_contiguous and p_out.size >= n
        if seg_p_out is not None:
            assert seg_p_out.dtype == np.float64 and seg_p_out.flags.c_contiguous and seg_p_out.size >= n
            want_seg_p = True
        seg_p = (seg_p_out if seg_p_out is not None else np.empty(max(n, 1), dtype=np.float64)) if want_seg_p else None
        n_seg = ctypes.c_int32(0)
        keep = []
        q = refine_params(criterion, threshold, n_cds, n_biopfams, average_threshold, edge_distance, trim, False, marker_ptr,
                          marker_id, keep)
        if degree is not None:
            assert degree.dtype == np.uint8 and degree.flags.c_contiguous and degree.size >= n
        adr = lambda a: None if a is None else a.ctypes.data  # noqa: E731
        _check(self._lib.gecco_crf_session_clusters_wire(
            self._h, adr(contig_ptr), nc, adr(gene_ptr), adr(degree), adr(attr_id) if attr16 is None else None, adr(attr16),
            adr(annotated), int(window), int(step), int(label), int(bool(pad)), ctypes.addressof(q), adr(p_out), adr(seg), cap,
            ctypes.addressof(n_seg), adr(seg_p) if want_seg_p else None, max(n, 1), adr(seg_off)))
        k = n_seg.value
        # (without seg_p the C side writes no offsets: the reused buffer would show those of an earlier call)
        res = (seg[:k].copy(), ((seg_p[: seg_off[k]] if seg_p_out is not None else seg_p[: seg_off[k]].copy()) if want_seg_p else None),
               (seg_off[: k + 1].copy() if want_seg_p else None),
               (p_out[:n] if p_out is not None else None))
        self._row_buffers = held  # (handed back only now: nobody else wrote into them meanwhile)
        return res


class Plan:
    """Resident batch: contig layout + model tables on one device; bulk arrays stay in
    caller-owned device memory (e.g. torch tensors) and launches go to the caller's stream."""

    def __init__(self, model: Model, contig_ptr, window, step=1, pad=True, device=0):
        self._lib = load_library()
        self.model = model
        contig_ptr = _i32(contig_ptr)
        h = _vp()
        _check(
            self._lib.gecco_crf_plan_create(
                model._h, int(device), _ptr(contig_ptr, _c_i32p), max(len(contig_ptr) - 1, 0), int(window), int(step),
                int(bool(pad)), ctypes.byref(h),
            )
        )
        self._h = h
        self.device = device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.gecco_crf_plan_free(h)

    @property
    def num_genes(self) -> int:
        return self._lib.gecco_crf_plan_num_genes(self._h)

    @property
    def num_windows(self) -> int:
        return self._lib.gecco_crf_plan_num_windows(self._h)

    @property
    def num_tiles(self) -> int:
        return self._lib.gecco_crf_plan_num_tiles(self._h)

    @property
    def tile_out(self) -> int:
        return self._lib.gecco_crf_plan_tile_out(self._h)

    @property
    def kernel_name(self) -> str:
        return self._lib.gecco_crf_plan_kernel_name(self._h).decode()

    def run_windowed(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, label=1, stream: int = 0):
        _check(self._lib.gecco_crf_plan_run_windowed(self._h, d_gene_ptr, d_attr_id, int(label), d_p_out, stream or None))

    def run_decode(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, d_y: int, label: int = 1, d_score: int = 0, stream: int = 0):
        """Windowed marginals + Viterbi labels in one pass over the CSR (shared state scores)."""
        _check(self._lib.gecco_crf_plan_run_decode(self._h, d_gene_ptr, d_attr_id or None, int(label), d_p_out, d_y,
                                                   d_score or None, stream or None))

    def run_decode_pipelined(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, prev: "Plan" = None, d_prev_y: int = 0, label: int = 1,
                             stream: int = 0):
        """Marginals of this plan's batch + Viterbi labels of the batch the previous call scored on `prev` (one launch
        when both qualify); `Plan.flush_decode_pipelined` delivers the labels of the last batch."""
        _check(self._lib.gecco_crf_plan_run_decode_pipelined(self._h, d_gene_ptr, d_attr_id or None, int(label), d_p_out,
                                                             prev._h if prev is not None else None, d_prev_y or None, stream or None))

    def bind_decode_pipelined(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, prev: "Plan" = None, d_prev_y: int = 0, label: int = 1,
                              stream: int = 0):
        """The same call with its arguments converted ONCE: returns a zero-argument callable for loops that repeat it on resident
        buffers (a launch-bound step is bound by what the host spends per call; the conversions of eight Python values are
        ~0.7 us of it)."""
        fn = self._lib.gecco_crf_plan_run_decode_pipelined
        args = (self._h, _vp(d_gene_ptr), _vp(d_attr_id or None), ctypes.c_int32(int(label)), _vp(d_p_out),
                prev._h if prev is not None else None, _vp(d_prev_y or None), _vp(stream or None))

        def call():
            rc = fn(*args)
            if rc:
                _check(rc)

        return call

    def flush_decode_pipelined(self, d_y: int, stream: int = 0):
        """Labels of the batch the last `run_decode_pipelined` call on this plan scored."""
        _check(self._lib.gecco_crf_plan_run_decode_pipelined(None, None, None, 1, None, self._h, d_y, stream or None))

    def run_marginals_full(self, d_gene_ptr: int, d_attr_id: int, d_marg: int, d_lognorm: int = 0, stream: int = 0):
        _check(self._lib.gecco_crf_plan_run_marginals_full(self._h, d_gene_ptr, d_attr_id, d_marg, d_lognorm or None, stream or None))

    def run_viterbi(self, d_gene_ptr: int, d_attr_id: int, d_y: int, d_score: int = 0, stream: int = 0):
        _check(self._lib.gecco_crf_plan_run_viterbi(self._h, d_gene_ptr, d_attr_id, d_y, d_score or None, stream or None))

    def run_segment(self, d_p: int, d_annotated: int, d_seg: int, max_seg: int, d_n_seg: int, threshold=0.8, n_cds=3,
                    edge_distance=0, trim=True, carry_state=False, stream: int = 0, criterion="gecco", n_biopfams=5,
                    average_threshold=0.6, d_marker_ptr: int = 0, d_marker_id: int = 0):
        """Cluster rows of device-resident probabilities, on the same stream as the marginals (no copies);
        `d_marker_ptr` / `d_marker_id`: device addresses of the genes' marker domains (antismash criterion)."""
        q = refine_params(criterion, threshold, n_cds, n_biopfams, average_threshold, edge_distance, trim, carry_state,
                          int(d_marker_ptr), int(d_marker_id))
        _check(self._lib.gecco_crf_plan_run_segment_ex(self._h, d_p, d_annotated, ctypes.byref(q), d_seg, int(max_seg), d_n_seg,
                                                       stream or None))

    def time_decode_pipelined(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, d_y: int, label=1, stream: int = 0, warmup=2,
                              iters=10) -> float:
        """ms per pipelined decode launch (this plan following itself), HIP events on `stream`."""
        ms = ctypes.c_float(0)
        _check(self._lib.gecco_crf_plan_time_decode_pipelined(self._h, d_gene_ptr, d_attr_id, int(label), d_p_out, d_y, stream or None,
                                                              int(warmup), int(iters), ctypes.byref(ms)))
        return ms.value

    def viterbi_stats(self, reset=True) -> dict:
        """What the 2-label Viterbi decoder met since the last reset: decisions inside the coarse margin (`candidates`),
        inside the margin where the difference form is not provably CRFsuite's (`inside_margin`), and the contigs / genes
        decoded again with CRFsuite's own recursion because of them.  Waits for the device."""
        out = (ctypes.c_int64 * 4)()
        _check(self._lib.gecco_crf_plan_viterbi_stats(self._h, out, int(bool(reset))))
        return {"candidates": int(out[0]), "inside_margin": int(out[1]), "contigs_redecoded": int(out[2]), "genes_redecoded": int(out[3])}

    def time_windowed(self, d_gene_ptr: int, d_attr_id: int, d_p_out: int, label=1, stream: int = 0, warmup=2, iters=10) -> float:
        ms = ctypes.c_float(0)
        _check(
            self._lib.gecco_crf_plan_time_windowed(
                self._h, d_gene_ptr, d_attr_id, int(label), d_p_out, stream or None, int(warmup), int(iters), ctypes.byref(ms)
            )
        )
        return ms.value
