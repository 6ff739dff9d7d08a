"""Pin the CPU oracle against every golden vector the reference holds for this path
(SURVEY.md §8c): tests/test_cli/data/BGC0001866.{features,genes,clusters}.tsv."""
import math
import os
import statistics

import numpy as np

from oracle import crf_oracle as orc
from oracle import lcrf
from tests.helpers import GOLDEN, golden_csr, read_tsv


def test_model_header_and_weights(oracle_model):
    m = oracle_model
    hdr = m["header"]
    assert hdr[0] == b"lCRF" and hdr[1] == 222468 and hdr[2] == b"FOMC" and hdr[3] == 100
    assert hdr[5] == 2 and hdr[6] == 2659
    assert hdr[7:] == (48, 84360, 86492, 184288, 184340)
    assert m["labels"] == ["0", "1"]
    assert len(m["attrs"]) == 2659 and m["attrs"][0] == "PF00750"
    assert m["n_feat"] == 4215
    assert m["state_mask"].sum() == 4211 and m["trans_mask"].sum() == 4
    np.testing.assert_array_equal(
        m["trans"], [[2.669891070463728, -2.599571900486168], [-2.6019205422130995, 2.5683226020688488]]
    )
    ai = m["attr_index"]
    assert m["state"][ai["PF00750"], 0] == 0.042198546846909164
    assert m["state"][ai["PF05746"], 0] == 0.9431860257345203
    assert m["state"][ai["PF13471"], 1] == 3.4651186873333413
    assert m["window_size"] == 20 and m["window_step"] == 1 and m["feature_type"] == "protein"


def test_md5_mismatch_raises(tmp_path):
    bad = tmp_path / "model.pkl.md5"
    bad.write_text("0" * 32)
    try:
        lcrf.load_pickle(os.path.join(GOLDEN, "model.pkl"), str(bad))
    except ValueError as e:
        assert "MD5 hash of model data does not match signature" in str(e)
    else:
        raise AssertionError("expected ValueError")


def test_windowed_marginals_match_genes_tsv(oracle_model):
    m = oracle_model
    ids, cptr, gptr, attr, expected, ann = golden_csr(m["attr_index"])
    assert len(ids) == 23 and len(attr) == 37 - (37 - len(attr))  # all 25 distinct domains known
    p = orc.windowed_marginals(m["state"], m["trans"], cptr, gptr, attr, W=20, step=1, label=1, pad=True)
    err = np.abs(p - expected).max()
    assert err <= 1e-15, err


def test_whole_contig_marginals_differ(oracle_model):
    """whole-contig marginals are different numbers."""
    m = oracle_model
    ids, cptr, gptr, attr, expected, ann = golden_csr(m["attr_index"])
    full, _ = orc.full_marginals(m["state"], m["trans"], cptr, gptr, attr)
    d = np.abs(full[:, 1] - expected).max()
    assert 1e-12 < d < 1e-9


def test_cluster_row(oracle_model):
    m = oracle_model
    ids, cptr, gptr, attr, expected, ann = golden_csr(m["attr_index"])
    p = orc.windowed_marginals(m["state"], m["trans"], cptr, gptr, attr, W=20)
    seg = orc.segment(p, ann, cptr, threshold=0.8, n_cds=3, edge_distance=0, trim=True)
    row = read_tsv(os.path.join(GOLDEN, "BGC0001866.clusters.tsv"))
    assert len(row) == 1 and len(seg) == 1
    c, number, a, b = seg[0]
    assert (c, number) == (0, 1)
    assert row[0]["cluster_id"] == "BGC0001866.1_cluster_1"
    genes = read_tsv(os.path.join(GOLDEN, "BGC0001866.genes.tsv"))
    assert min(int(g["start"]) for g in genes[a:b]) == int(row[0]["start"])
    assert max(int(g["end"]) for g in genes[a:b]) == int(row[0]["end"])
    assert set(ids[a:b]) == set(row[0]["proteins"].split(";"))
    # Cluster.average_probability uses statistics.mean (gecco/model.py:442-447)
    assert abs(statistics.mean(p[a:b].tolist()) - float(row[0]["average_p"])) <= 2e-16
    assert abs(max(p[a:b]) - float(row[0]["max_p"])) <= 1e-15
    feats = read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv"))
    doms = sorted({r["domain"] for r in feats if r["protein_id"] in set(ids[a:b])})
    assert doms == row[0]["domains"].split(";")


def test_features_tsv_cluster_probability(oracle_model):
    """features.tsv repeats the gene's p on every domain row (features.py:92-96)."""
    m = oracle_model
    ids, cptr, gptr, attr, expected, ann = golden_csr(m["attr_index"])
    p = dict(zip(ids, orc.windowed_marginals(m["state"], m["trans"], cptr, gptr, attr, W=20)))
    for r in read_tsv(os.path.join(GOLDEN, "BGC0001866.features.tsv")):
        assert abs(p[r["protein_id"]] - float(r["cluster_probability"])) <= 1e-15


def test_viterbi_all_ones(oracle_model):
    m = oracle_model
    ids, cptr, gptr, attr, expected, ann = golden_csr(m["attr_index"])
    lab, sc = orc.viterbi(m["state"], m["trans"], cptr, gptr, attr)
    assert lab.tolist() == [1] * 23 and math.isfinite(sc[0])
