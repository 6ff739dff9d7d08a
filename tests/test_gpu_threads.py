"""The C ABI under concurrent callers (SURVEY.md 8b "Threading": the model handle is immutable after load, scratch is
per call / per session).  ctypes releases the GIL for the duration of a call, so these threads really do run the
library concurrently: one-shot entry points on the model's own session, private sessions, the segmenter's per-thread
scratch, resident plans on their own streams and the columnar packer's worker pool, all at once."""
import threading

import numpy as np
import pytest

from tests.helpers import synth_contigs

pytestmark = pytest.mark.gpu

TOL = 1e-12


def test_concurrent_callers_get_their_own_results(oracle_model):
    import os

    from gecco_amd import _native as nat
    from gecco_amd import tables
    from oracle import crf_oracle as orc
    from oracle import lcrf
    from tests.helpers import GOLDEN

    model = nat.Model.from_lcrf(lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"))["blob"])
    A = oracle_model["state"].shape[0]
    n_threads, rounds = 6, 4
    cases = []
    for t in range(n_threads):
        rng = np.random.default_rng(100 + t)
        lengths = list(rng.integers(1, 300, size=40 + 10 * t)) + [19, 20, 21, 2500]
        cptr, gptr, attr = synth_contigs(rng, lengths, A)
        p = orc.windowed_marginals(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr, 20, 1, 1, True)
        y = orc.viterbi(oracle_model["state"], oracle_model["trans"], cptr, gptr, attr)[0]
        ann = (np.diff(gptr) > 0).astype(np.uint8)
        v = np.sort(p)
        k = len(v) // 2
        while v[k + 1] - v[k] < 1e-6:
            k += 1
        thr = float(0.5 * (v[k] + v[k + 1]))
        seg = orc.segment(p, ann, cptr, thr, 2, 0, True)
        cases.append((cptr, gptr, attr, ann, p, y, thr, seg))
    S = tables.StringColumn.from_sequence
    errors = []
    barrier = threading.Barrier(n_threads)

    def work(t):
        try:
            cptr, gptr, attr, ann, p_exp, y_exp, thr, seg_exp = cases[t]
            ses = nat.Session(model, [0])
            ses.set_chunk_genes(1024 + 512 * t)
            barrier.wait()
            for _ in range(rounds):
                p = model.windowed_marginals(cptr, gptr, attr, 20, 1, 1, True)          # shared default session
                assert np.abs(p - p_exp).max() <= TOL
                p2, y2 = ses.decode(cptr, gptr, attr, 20, 1, 1, True)                     # private session
                assert np.abs(p2 - p_exp).max() <= TOL and np.array_equal(y2.astype(np.int32), y_exp)
                seg, _, _, _ = ses.clusters(cptr, gptr, attr, ann, 20, 1, 1, True, thr, 2, 0, True)
                assert seg.tolist() == seg_exp.tolist()
                assert nat.segment(p_exp, ann, cptr, thr, 2, 0, True).tolist() == seg_exp.tolist()   # per-thread scratch
                y3, _ = model.viterbi(cptr, gptr, attr)
                assert np.array_equal(y3.astype(np.int32), y_exp)
                # the packer's worker pool serialises its jobs
                n = 2000 + 100 * t
                pid = [f"t{t}_g{i}" for i in range(n)]
                pk = nat.PackedTables(model, S(["c"] * n), S(pid), np.arange(n, dtype=np.int64), S(["PF00109"] * n),
                                      np.ones(n, dtype=np.int64), S(["c"] * n), S(pid), np.arange(n, dtype=np.int64))
                assert pk.n_genes == n and pk.nnz == n and pk.contig_ptr.tolist() == [0, n]
        except BaseException as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a caller is stuck"
    assert not errors, errors
