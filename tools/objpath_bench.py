"""Host-only timing of ClusterCRF.predict_probabilities' object loops (no device): the steps of gecco_amd/crf.py around the
batch driver call, on the object model of gecco_amd.model, with the scores replaced by a constant array.
    python tools/objpath_bench.py [n_contigs] [genes_per_contig]"""
import gc
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from gecco_amd import packing  # noqa: E402
from gecco_amd.crf import ClusterCRF  # noqa: E402
from gecco_amd.model import Domain, Gene, Protein, Source, Strand  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def build(attrs, n_contigs, per, seed=0):
    rng = np.random.default_rng(seed)
    genes = []
    for c in range(n_contigs):
        src = Source(f"contig_{c:04d}")
        for i in range(per):
            k = int(rng.integers(0, 4))
            doms = [Domain(attrs[a], 10 * j + 1, 10 * j + 9, "Pfam", 1e-10, 1e-12) for j, a in enumerate(rng.integers(0, len(attrs), size=k))]
            genes.append(Gene(src, 1000 * i, 1000 * i + 900, Strand.Coding, Protein(f"c{c:04d}_{i}", None, doms)))
    return genes


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    crf = ClusterCRF.trained(GOLDEN)
    genes = build(crf.model.attributes_, nc, per)
    n = len(genes)

    def fake_score(batch, W, step, label, pad, progress, total):
        return np.full(int(batch.item_ptr[-1]), 0.5)

    crf._score = fake_score
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        crf.predict_probabilities(genes[:2000])
        best = 1e9
        for _ in range(5):
            gc.collect()
            t0 = time.perf_counter()
            out = crf.predict_probabilities(genes)
            best = min(best, time.perf_counter() - t0)
            del out
    print(f"{n} genes: predict_probabilities without the device call {best * 1e6 / n:.3f} us per gene = {n / best / 1e6:.2f} M genes/s")
    steps = getattr(crf, "_host_steps", None)
    if steps:
        print({k: round(v * 1e6 / n, 3) for k, v in steps.items()})


if __name__ == "__main__":
    main()
