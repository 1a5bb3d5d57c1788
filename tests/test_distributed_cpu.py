"""The N > 1 path on CPU: two gloo ranks shard one batch by sample count, synthesise their shards
(the oracle stands in for the GPU kernel here -- this test is about the host logic), and the
reductions bench.py uses give the whole-job numbers.  No data-path collective exists to test."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from nvspeechplayer_amd import workloads
    from nvspeechplayer_amd.sharding import reduce_throughput, shard_bounds
    from tests import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = workloads.make("cfg3", 24)
    bounds = shard_bounds(batch.sample_counts(), world)
    mine = batch.slice(int(bounds[rank]), int(bounds[rank + 1] - bounds[rank]))
    pcm, out_start, total = oracle.batch_synthesize(batch["sr"], mine, threads=1)
    elapsed, samples = reduce_throughput(1.0 + rank, total, dist)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), pcm=pcm, bounds=bounds, elapsed=elapsed, samples=samples,
             total=total)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_sharding_gloo(tmp_path):
    import torch.multiprocessing as mp
    from nvspeechplayer_amd import workloads
    from tests import oracle
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    batch = workloads.make("cfg3", 24)
    full, _, total = oracle.batch_synthesize(batch["sr"], batch, threads=2)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([p["pcm"] for p in parts]), full)       # shards are disjoint and cover
    for p in parts:
        assert float(p["elapsed"]) == 2.0                                        # max over ranks
        assert float(p["samples"]) == float(total)                               # sum over ranks
    assert int(parts[0]["total"]) + int(parts[1]["total"]) == total


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two rank processes, they rendezvous (gloo here),
    deal the node's batch (2 x the per-GPU configuration) into two shards of near-equal SAMPLE count, build their shards and
    reduce; rank 0's JSON line comes back through the parent.  --dry-run leaves out only the GPU launches, so this runs in the
    CPU container; tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu runs the same command for real."""
    import json
    import subprocess
    from nvspeechplayer_amd import workloads
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for workload, per_gpu in (("cfg2", 300), ("cfg4", 1100)):
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", workload,
                                       "--utterances", str(per_gpu), "--steps", "3", "--warmup", "1"], env=env, cwd=str(tmp_path), timeout=280)
        lines = [l for l in out.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1, out
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dry_run"] is True
        c = d["config"]
        assert c["world_size"] == 2 and c["process_group"] == "gloo"
        assert c["node_utterances"] == 2 * per_gpu
        counts = workloads.sample_counts(workload, 2 * per_gpu)
        assert c["node_samples"] == int(counts.sum()) == int(d["total_samples_all_ranks"])        # the shards cover the node batch
        b = c["shard_bounds"]
        assert b[0] == 0 and b[2] == 2 * per_gpu and c["utterances_per_gpu"] == b[1]
        halves = [int(counts[:b[1]].sum()), int(counts[b[1]:].sum())]
        assert abs(halves[0] - halves[1]) <= 2 * int(counts.max())                                # balanced by samples, not by count
        assert c["samples_per_gpu"] == halves[0]
    # without --utterances the N > 1 run also carries BASELINE's node-wide configurations at their node sizes as extra keys
    # (--dry-run: the deal alone, from the closed-form lengths)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"],
                                  env=env, cwd=str(tmp_path), timeout=280)
    d = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][0])
    assert d["config"]["node_utterances"] == 2 * 65536 and d["config"]["rccl_ranks"] is None      # gloo here; RCCL reports its ranks
    for key, per_gpu, wl in (("cfg3_node", 125000, "cfg3"), ("cfg4_node", 32 * 16384, "cfg4")):
        e = d[key]
        counts = workloads.sample_counts(wl, 2 * per_gpu)
        assert e["node_utterances"] == 2 * per_gpu and e["node_samples"] == int(counts.sum()) == int(e["total_samples_all_ranks"])
        b = e["shard_bounds"]
        assert b[0] == 0 and b[2] == 2 * per_gpu
        assert abs(int(counts[:b[1]].sum()) - int(counts[b[1]:].sum())) <= 2 * int(counts.max())
        assert e["value"] is None and e["launches"] == 5


@pytest.mark.timeout(900)
def test_bench_eight_ranks_dry_run(tmp_path):
    """The run the driver makes on a whole node -- `bench.py --gpus 8` -- with everything but the GPU launches (--dry-run): eight rank
    processes rendezvous, the node's batch (8 x 65 536 utterances) is dealt into eight contiguous shards whose SAMPLE counts agree
    within 0.1 %, every rank builds its own shard, the reductions give the node's totals; BASELINE configs[3] (10^6 utterances) and
    configs[4] (256 variants x 16 384) are dealt at their node sizes the same way.  No rank may need more than 4 GB of host memory
    (eight of them share a node's RAM, and the driver's box has a quota)."""
    import json
    import resource
    import subprocess
    from nvspeechplayer_amd import workloads
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "0"],
                                  env=env, cwd=str(tmp_path), timeout=850)
    peak_kb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss      # the largest of the processes waited for: bench.py's ranks among them
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["scaling"] == "weak" and c["world_size"] == 8 and c["process_group"] == "gloo"
    assert c["node_utterances"] == 8 * 65536

    def balanced(bounds, counts, what):
        assert len(bounds) == 9 and bounds[0] == 0 and bounds[-1] == len(counts) and all(b1 > b0 for b0, b1 in zip(bounds, bounds[1:]))
        csum = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
        shares = np.diff(csum[np.asarray(bounds)])
        assert int(shares.sum()) == int(counts.sum())
        spread = float(shares.max() - shares.min()) / float(shares.mean())
        assert spread < 1e-3, "%s: shards differ by %.3f %% of their samples" % (what, 100 * spread)
        return shares
    counts = workloads.sample_counts("cfg2", 8 * 65536)
    shares = balanced(c["shard_bounds"], counts, "cfg2 x 8")
    assert c["node_samples"] == int(counts.sum()) == int(d["total_samples_all_ranks"]) and c["samples_per_gpu"] == int(shares[0])
    for key, per_gpu, wl in (("cfg3_node", 125000, "cfg3"), ("cfg4_node", 32 * 16384, "cfg4")):
        e = d[key]
        counts = workloads.sample_counts(wl, 8 * per_gpu)
        assert e["node_utterances"] == 8 * per_gpu and e["node_samples"] == int(counts.sum()) == int(e["total_samples_all_ranks"])
        balanced(e["shard_bounds"], counts, key)
    assert peak_kb < 4 * 1024 * 1024 or peak_kb == before, "a rank's peak RSS was %.1f GB" % (peak_kb / 1048576.0)


def test_sorted_deal_balances_a_batch_whose_long_utterances_cluster():
    """utterances sorted by length, blocks of 64 dealt round-robin.  On a NON-periodic batch -- lengths log-uniform over
    two decades, the long ones clustered at the front -- the contiguous deal (near-equal sample totals) gives one rank a few very long
    utterances: its launch lasts as long as its longest wavefront, and its wavefront-time (sum over wavefronts of the longest lane) is
    far from the others'.  Under the sorted deal every rank sees the same length distribution: longest wavefronts within 5 %, samples
    and wavefront-times within 2 %; every utterance is dealt exactly once."""
    from nvspeechplayer_amd.sharding import longest_wave, shard_deal, wave_time
    rng = np.random.default_rng(11)
    n, world = 200000, 8
    counts = np.exp(rng.uniform(np.log(2000.0), np.log(200000.0), n)).astype(np.int64)
    # the long ones at the front, in no particular order among themselves
    key = -(counts // 20000) * 10.0 + rng.random(n)
    counts = counts[np.argsort(key, kind="stable")]
    for deal in ("contiguous", "sorted"):
        parts = shard_deal(counts, world, deal)
        allu = np.sort(np.concatenate(parts))
        assert np.array_equal(allu, np.arange(n))                                           # a partition
        longest = np.array([longest_wave(counts, m) for m in parts], dtype=np.float64)
        samples = np.array([counts[m].sum() for m in parts], dtype=np.float64)
        wtime = np.array([wave_time(counts, m) for m in parts], dtype=np.float64)
        if deal == "contiguous":
            assert samples.max() / samples.mean() < 1.01                                     # what it balances
            assert longest.max() / longest.mean() > 1.5                                      # what it does not
            lens0 = np.sort(counts[parts[0]]); lens7 = np.sort(counts[parts[-1]])
            assert np.median(lens0) > 8 * np.median(lens7)                                   # rank 0: few long utterances; rank 7: many short ones
        else:
            assert longest.max() / longest.mean() <= 1.05
            assert samples.max() / samples.mean() < 1.02 and wtime.max() / wtime.mean() < 1.02
    # the engine's node object deals the same way under its option "deal" (tests/test_gpu_parity.py runs it on the GPU)
