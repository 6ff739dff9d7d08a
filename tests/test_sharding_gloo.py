"""N > 1 path on CPU: two gloo ranks each score their own contig shard (the ORACLE stands in
for the engine here -- there is no GPU in this container) and rank 0 reassembles per-gene
results; the result must equal the unsharded computation exactly.  No data-path collective:
only the final gather of results uses torch.distributed."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from gecco_amd import sharding, synth
    from oracle import crf_oracle as orc

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    w, trans = synth.synth_model(300, rng)
    lengths = list(rng.integers(1, 150, size=37))
    cptr, gptr, attr = synth.synth_contigs(rng, lengths, 300)
    shards = sharding.partition_contigs(np.diff(cptr), world)
    scp, sgp, sat, gidx = sharding.extract_shard(cptr, gptr, attr, shards[rank])
    p = orc.windowed_marginals(w, trans, scp, sgp, sat, 20, 1, 1, True)
    gathered = [None] * world
    dist.all_gather_object(gathered, (gidx, p))
    if rank == 0:
        full = sharding.scatter_results(int(cptr[-1]), gathered)
        ref = orc.windowed_marginals(w, trans, cptr, gptr, attr, 20, 1, 1, True)
        q.put((np.array_equal(full, ref), [len(s) for s in shards], [int(np.diff(cptr)[s].sum()) for s in shards]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, counts, loads = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
    assert sum(counts) == 37 and abs(loads[0] - loads[1]) <= 150  # balanced by gene count


def test_partition_and_extract_roundtrip():
    from gecco_amd import sharding, synth

    rng = np.random.default_rng(1)
    cptr, gptr, attr = synth.synth_contigs(rng, [5, 1, 80, 33, 2, 64, 17], 50)
    shards = sharding.partition_contigs(np.diff(cptr), 3)
    assert sorted(np.concatenate(shards).tolist()) == list(range(7))
    seen = []
    for s in shards:
        scp, sgp, sat, gidx = sharding.extract_shard(cptr, gptr, attr, s)
        assert scp[-1] == len(gidx) and sgp[-1] == len(sat)
        for k, g in enumerate(gidx):
            assert sat[sgp[k]:sgp[k + 1]].tolist() == attr[gptr[g]:gptr[g + 1]].tolist()
        seen += gidx.tolist()
    assert sorted(seen) == list(range(int(cptr[-1])))
