"""bench.py -- throughput of the Klatt hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg1|cfg3|cfg4] [--mode 0]

A step is one pass of the synthesis kernel over one batch of synthetic frame streams that is already resident in
HBM (frames in, int16 PCM out, both in HBM).  The headline workload is BASELINE.json configs[2], the largest
single-GPU configuration: 65 536 IPA utterances from sampleIpa.txt through the frame producer (1.5e9 samples per
step); the other configurations are selectable and cfg1 (configs[1]) is measured as an extra key.

N > 1: one process per GPU.  Under `torch.distributed.run` the ranks come from the environment; a plain
`python bench.py --gpus N` starts its own N rank processes (the parent never touches the GPU, relays rank 0's JSON
line and returns the ranks' worst exit code).  The node's batch is N times the per-GPU configuration (weak scaling);
its utterance list is cut into N contiguous shards of near-equal total SAMPLE count (closed-form lengths,
nvspeechplayer_amd.sharding.shard_bounds) and every rank builds and synthesises its own shard.  Utterances are
independent: there is no collective on the data path; the process group only carries the timing barrier, the max of
the elapsed times and the sum of the sample counts.

The K timed launches are individually bracketed by HIP events on the engine's stream (speechPlayer_batch_time); their
mean is the kernel duration the roofline uses, the wall clock around all K (barrier + device synchronize on both
sides, max over ranks) gives `value`.  The CPU baseline is the oracle (oracle/klatt_oracle.c, "port") timed on this
box's host cores on a bounded sample of the same workload, rank 0 at N = 1 only.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue: 256 CUs x 4 SIMDs; a wave64 f64 instruction occupies its SIMD for 4 cycles (78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 flop
# x 2.4 GHz), any other VALU instruction for 2 (profiles/r1_ubench_issue_rates.txt: ~2.0 / ~1.1 ns at the clock the chip holds under
# this load, 2.06-2.07 GHz -- GRBM_GUI_ACTIVE / 8 XCDs / duration, profiles/r6_pmc.json "held_clock_hz")
SIMDS, F64_ISSUE_CYCLES, OTHER_ISSUE_CYCLES, CLOCK_HZ_HELD_DEFAULT = 1024, 4.0, 2.0, 2.07e9
PMC_FILE = os.path.join(ROOT, "profiles", "r6_pmc.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed launches (cfg2: ~12 ms each, so the default times > 1 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--utterances", type=int, default=0, help="override the batch size per GPU")
    ap.add_argument("--mode", type=int, default=0, help="arithmetic mode (include/speechPlayer_batch.h)")
    ap.add_argument("--layout", type=int, default=-1, help="-1: engine's choice, 1: stage-parallel workgroups, 0: one wavefront per 64 utterances")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-large-batch", dest="no_extras", action="store_true",
                    help="skip the extra measurements (MODE_FAST, cfg1, cfg1 recipe at 65536 utterances)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall time to spend on the CPU baseline")
    ap.add_argument("--dry-run", action="store_true",
                    help="everything but the GPU: ranks, rendezvous (gloo), shard dealing, workload build, reductions; value is null")
    return ap.parse_args(argv)


def engine_source_digest():
    """Identifies the kernels a PMC file was collected from: sha1 over the kernel sources csrc/klatt_*.h and klatt_engine.hip
    and the compiler flags they are built with."""
    from nvspeechplayer_amd import _native
    csrc = os.path.join(ROOT, "nvspeechplayer_amd", "csrc")
    h = hashlib.sha1(" ".join(_native.HIPCC_FLAGS).encode())
    for f in sorted(os.listdir(csrc)):
        if f.startswith("klatt_") and f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_name(info):
    """The kernel of the batch's largest group (speechPlayer_batch_kernelInfo)."""
    if info.get("lane_pipelined"):
        return "klatt_lanepipe (cascade across lanes, %d-sample hand-overs)" % info["stage_parallel_chunk"]
    if info.get("direct"):
        return "klatt_direct (8 stages, coefficients computed in place from %d MB of per-frame seeds, %d-sample hand-overs)" % (info.get("direct_mbytes", 0), info["stage_parallel_chunk"])
    if info["stage_parallel_chunk"]:
        return "klatt_systolic (stage-parallel%s%s%s, %d-sample hand-overs)" % (
            ", noisy" if info.get("noisy_group") else "", ", nasal-free" if info.get("nasal_free") else "",
            ", flat stages: parameters and coefficients from %d tracks (klatt_tracks, %d MB)" % (info["tracks"], info["track_mbytes"]) if info.get("tracked") else "",
            info["stage_parallel_chunk"])
    return "klatt_synthesize (lane kernel)"


def pipeline_extra(device, mode, layout, n_batches=8, players=4, workers=3, copy_out=False, pinned=True, source="frames"):
    """Sustained end-to-end throughput: `n_batches` DISTINCT cfg2-sized batches (other noise seeds, other pitches: nothing of one batch
    is reused by the next) through `players` BatchPlayers; `workers` host threads run the set call of the batches to come while the GPU
    synthesises the current one.  source = "frames": the caller hands over 376-byte frames, one list per utterance
    (speechPlayer_batch_setUtterances: classification, lane packing, track planning, 0.6 GB uploaded per batch; `pinned`: the frames are
    in page-locked memory the library handed out, else pageable numpy arrays); source = "ipa": the caller hands over IPA text, base
    pitches and seeds (speechPlayer_batch_setIpa: the native producer builds the distinct frame lists as 32-byte records, the device
    expands them, utterances share their list's frames) -- the producer's work is INSIDE the clock.  PCM stays in HBM, or (copy_out) is
    delivered into one host buffer per player: the device puts it into dense utterance order, one copy per batch crosses the link
    (speechPlayer_batch_readAllAsync) while the next batch is synthesised."""
    import threading
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer, host_array, workloads
    batches = []
    for k in range(n_batches):
        if source == "ipa":
            spec = workloads.cfg2_spec(65536, first=k * 65536)
            spec["basePitch"] = spec["basePitch"] * (1.0 + 0.01 * k)
            batches.append(spec)
            continue
        b = workloads.make("cfg2", 65536, first=k * 65536)
        fr = host_array(b["frames"].shape, np.float64) if pinned else np.empty_like(b["frames"])
        fr[...] = b["frames"]
        fr[:, 0] *= 1.0 + 0.01 * k; fr[:, 46] *= 1.0 + 0.01 * k
        b["frames"] = fr
        batches.append(b)
    sr = workloads.SR

    def set_batch(bp, b):
        if source == "ipa":
            bp.setIpa(**b)
        else:
            bp.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
    bps = [BatchPlayer(sr, device=device, mode=mode, layout=layout) for _ in range(players)]
    ready = [threading.Event() for _ in range(n_batches)]
    free = [threading.Semaphore(1) for _ in range(players)]
    set_s = [0.0] * n_batches
    err = []
    nxt = [0]
    lock = threading.Lock()

    def work():
        try:
            while True:
                with lock:
                    k = nxt[0]; nxt[0] += 1
                if k >= n_batches:
                    return
                free[k % players].acquire()          # the player's previous batch has been synthesised (and read)
                t = time.perf_counter()
                set_batch(bps[k % players], batches[k])
                set_s[k] = time.perf_counter() - t
                ready[k].set()
        except Exception as e:      # noqa: BLE001
            err.append(e)
            for r in ready:
                r.set()
    # warm-up: one batch through every player (allocations, first launches)
    for i, bp in enumerate(bps):
        set_batch(bp, batches[i % n_batches])
        bp.synthesize()
    outs = None
    if copy_out:
        n_out = int(max(bp.totalSamples for bp in bps) * 1.05)
        outs = [host_array(n_out, np.int16) if pinned else np.empty(n_out, dtype=np.int16) for _ in range(players)]
        for i, o in enumerate(outs):
            o[...] = 0
            bps[i].readAll(out=o)
    total = 0
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work) for _ in range(workers)]
    for th in ths:
        th.start()
    synth_s = 0.0
    in_flight = []                     # batches launched and not yet retired: one stays in flight while the next is launched

    def retire(j):
        bpj = bps[j % players]
        bpj.wait()
        if copy_out and pinned:
            bpj.readWait()
        elif copy_out:
            bpj.readAll(out=outs[j % players])
        n = bpj.totalSamples
        free[j % players].release()
        return n
    for k in range(n_batches):
        ready[k].wait()
        if err:
            raise err[0]
        bp = bps[k % players]
        t = time.perf_counter()
        bp.synthesize(wait=False)
        if copy_out and pinned:
            bp.readAllAsync(outs[k % players])
        in_flight.append(k)
        while len(in_flight) > 1:
            total += retire(in_flight.pop(0))
        synth_s += time.perf_counter() - t
    t = time.perf_counter()
    while in_flight:
        total += retire(in_flight.pop(0))
    synth_s += time.perf_counter() - t
    elapsed = time.perf_counter() - t0
    for th in ths:
        th.join()
    for bp in bps:
        bp.close()
    return {"value": total / elapsed, "unit": "samples/s", "batches": n_batches, "players": players, "setter_threads": workers,
            "planner_threads_per_set_call": int(os.environ.get("SPEECHPLAYER_PLAN_THREADS", "8")), "host_cores": usable_cores(),
            "input": ("IPA text + base pitch + seed per utterance (speechPlayer_batch_setIpa: producer inside the clock, 32-byte records, shared frame lists)"
                      if source == "ipa" else "376-byte frames, one list per utterance (speechPlayer_batch_setUtterances)"),
            "host_buffers": "page-locked (speechPlayer_hostAlloc): frames in and PCM out are one DMA each" if pinned else "pageable numpy arrays",
            "pcm": ("one dense copy per batch into a host buffer per player (speechPlayer_batch_readAllAsync), beside the next batch's synthesis" if pinned
                    else "copied to one host buffer per player (speechPlayer_batch_readAll)") if copy_out else "left in HBM",
            "elapsed_s": round(elapsed, 3), "set_call_s_mean_under_load": round(float(np.mean(set_s)), 4), "gpu_side_s_per_batch": round(synth_s / n_batches, 4),
            "samples_per_batch": total // n_batches}


def single_stream_extra(pulls=50, pull=8192):
    """The reference's own use (nvdaAddon/synthDrivers/nvSpeechPlayer/__init__.py:62-81): ONE handle, pulled 8192 samples at a time.
    Median milliseconds per speechPlayer_synthesize call, beside the same pulls of the oracle on one host core."""
    import numpy as np
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads
    from tests import oracle
    b = workloads.make("cfg2", 8)
    rows = [(b["frames"][k], int(b["min"][k]), int(b["fade"][k]), bool(b["isnull"][k])) for k in range(int(b["frame_start"][8]))]
    need = pulls * pull + pull
    p = eng.SpeechPlayer(22050, noiseSeed=1)
    o = oracle.OraclePlayer(22050, seed=1)
    queued = 0
    while queued < need:                     # the eight sentences again and again: no pull runs dry
        for fr, m, f, nul in rows:
            p.queueFrameSamples(None if nul else eng.Frame.from_array(fr), m, f)
            o.queue(None if nul else fr, m, f)
            queued += max(m, max(f, 1) + 1) + 1
    p.synthesize(pull); o.synthesize(pull)   # first call: allocations, first launch
    t_gpu, t_cpu, same = [], [], True
    for _ in range(pulls):
        t = time.perf_counter(); buf = p.synthesize(pull); t_gpu.append(time.perf_counter() - t)
        t = time.perf_counter(); exp = o.synthesize(pull); t_cpu.append(time.perf_counter() - t)
        same = same and buf is not None and buf.length == pull and np.array_equal(np.frombuffer(buf, dtype=np.int16)[:pull], exp)
    p.close()
    # the same pulls of a MODE_FAST handle (speechPlayer_setGlobalOption("live_mode", 1): fused multiply-adds in the filters)
    from nvspeechplayer_amd import _native
    L = _native.load()
    t_fast = []
    try:
        if L.speechPlayer_setGlobalOption(b"live_mode", 1) == 0:
            pf = eng.SpeechPlayer(22050, noiseSeed=1)
            queued = 0
            while queued < need:
                for fr, m, f, nul in rows:
                    pf.queueFrameSamples(None if nul else eng.Frame.from_array(fr), m, f)
                    queued += max(m, max(f, 1) + 1) + 1
            pf.synthesize(pull)
            for _ in range(pulls):
                t = time.perf_counter(); pf.synthesize(pull); t_fast.append(time.perf_counter() - t)
            pf.close()
    finally:
        L.speechPlayer_setGlobalOption(b"live_mode", 0)
    return {"what": "one live handle, %d pulls of %d samples (speechPlayer_synthesize); the same pulls of the oracle on one host core" % (pulls, pull),
            "ms_per_pull_median": float(np.median(t_gpu)) * 1e3, "ms_per_pull_min": float(np.min(t_gpu)) * 1e3,
            "mode_fast_ms_per_pull_median": float(np.median(t_fast)) * 1e3 if t_fast else None,
            "oracle_ms_per_pull_median": float(np.median(t_cpu)) * 1e3, "real_time_ms_per_pull": pull / 22050.0 * 1e3, "pcm_equal": bool(same)}


def live_handles_extra(n=1024, pull=8192, pulls=4):
    """Many live handles advanced together (speechPlayer_synthesizeMany, PCM left in HBM): kernel ms per 8192-sample pull when the
    handles speak IN STEP (one sentence, same sample: a shared wavefront's chunks are steady or fading as a whole) and when they are
    UNRELATED (the eight sampleIpa sentences, every handle skewed by a pull of its own first: every chunk of a shared wavefront runs
    sample by sample) -- with a wavefront per handle (the default up to 1536 handles, option "live_alone") and with 64 handles per
    wavefront ("live_alone" 1: what larger pulls get; up to 16 384 handles such a pull lasts as long as one workgroup's)."""
    import numpy as np
    import nvspeechplayer_amd as eng
    from nvspeechplayer_amd import workloads, _native
    L = _native.load()
    b = workloads.make("cfg2", 8)
    fs = b["frame_start"]
    lines = [[(b["frames"][k], int(b["min"][k]), int(b["fade"][k]), bool(b["isnull"][k])) for k in range(int(fs[u]), int(fs[u + 1]))] for u in range(8)]
    out = {"handles": n, "samples_per_pull": pull, "what": "kernel ms per pull of all handles together (speechPlayer_lastLiveKernelMs), median of %d" % pulls}
    try:
        for key, unrelated in (("in_step", False), ("unrelated", True)):
            out[key] = {}
            for policy, alone in (("a_wavefront_per_handle", 1536), ("64_handles_per_wavefront", 1)):
                assert L.speechPlayer_setGlobalOption(b"live_alone", alone) == 0
                rng = np.random.default_rng(3)
                players = [eng.SpeechPlayer(22050, noiseSeed=k) for k in range(n)]
                for k, p in enumerate(players):
                    rows = lines[k % 8] if unrelated else lines[5]
                    queued = 0
                    while queued < (pulls + 2) * pull + 4096:
                        for fr, m, f, nul in rows:
                            p.queueFrameSamples(None if nul else eng.Frame.from_array(fr), m, f)
                            queued += max(m, max(f, 1) + 1) + 1
                    if unrelated:
                        p.synthesize(int(rng.integers(1, 4000)))
                group = eng.LiveGroup(players)
                group.pullDevice(64)
                kms = []
                for _ in range(pulls):
                    _, _, produced = group.pullDevice(pull)
                    kms.append(float(L.speechPlayer_lastLiveKernelMs(0)))
                ms = float(np.median(kms))
                out[key][policy] = {"kernel_ms": ms, "samples_per_s": float(produced.sum()) / (ms * 1e-3), "all_handles_full": bool((produced == pull).all())}
                for p in players:
                    p.close()
                del group, players
    finally:
        L.speechPlayer_setGlobalOption(b"live_alone", 1536)
    return out


def cfg0_cpu_extra():
    """BASELINE configs[0]: a single steady /a/ at 120 Hz, 1 s at 22.05 kHz, on the CPU (the oracle on one core; SURVEY 8c's recipe,
    test_playVowelchart path): samples/s and multiples of real time."""
    import numpy as np
    from tests import oracle, scenarios
    ref = scenarios.Ref()
    fr = scenarios.vowel_frame(ref, "a", 120.0)
    best = None
    for _ in range(20):
        o = oracle.OraclePlayer(22050, seed=0)
        o.queue(fr, 22050, 1102)
        t = time.perf_counter()
        pcm = o.synthesize(22050)
        dt = time.perf_counter() - t
        o.close()
        best = dt if best is None else min(best, dt)
    return {"workload": "BASELINE configs[0]: single steady /a/ 120 Hz, 1 s at 22.05 kHz, oracle on one host core (best of 20)",
            "value": len(pcm) / best, "unit": "samples/s", "seconds": best, "times_real_time": len(pcm) / best / 22050.0, "cores": 1, "kind": "port"}


def usable_cores():
    """Host threads this process may really use: CPU affinity, capped by the cgroup CPU quota
    (on the GPU box 256 hardware threads are visible but the container's quota is 16 CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(batch, target_seconds):
    """Oracle on the host cores: one OpenMP thread per core, each synthesising whole utterances of the same workload (no
    shared state: every utterance has its own player and noise stream).  Bounded sample: a leading slice of the workload
    sized from a one-thread probe so that one pass takes about `target_seconds` (the headline workload is 1.5e9 samples
    -- minutes of CPU work -- so the whole of it is not run)."""
    from tests import oracle
    cores = usable_cores()
    probe = batch.slice(0, min(8, batch.n_utt))
    t0 = time.perf_counter()
    _, _, total = oracle.batch_synthesize(batch["sr"], probe, threads=1)
    one_core = total / (time.perf_counter() - t0)
    budget = one_core * cores * target_seconds * 0.8                 # samples the cores should manage in the time
    mean = max(1.0, float(batch.sample_counts().mean()))
    n = int(min(batch.n_utt, max(cores * 8, budget / mean)))
    n -= n % 8 if n >= 16 else 0                                      # whole periods of the eight sampleIpa lines
    sample = batch.slice(0, n)
    oracle.batch_synthesize(batch["sr"], batch.slice(0, min(batch.n_utt, cores)), threads=cores)   # start the thread pool
    done, reps, t0 = 0, 0, time.perf_counter()
    while True:
        _, _, total = oracle.batch_synthesize(batch["sr"], sample, threads=cores)
        done += total
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds * 0.75 or reps >= 200:
            break
    return {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "the first %d of %d utterances x %d pass(es) = %d samples on %d OpenMP threads in %.1f s; "
                      "1 thread: %.3g samples/s" % (n, batch.n_utt, reps, done, cores, dt, one_core)}


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the ranks itself
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relay rank 0's
    output, return the worst exit code.  Nothing here touches HIP or torch: the children are fresh interpreters."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's output is read on a thread so that the loop below can watch every rank: if one dies, the others would wait
    # in a barrier for ever -- they are ended (by PID: these are our own children) and the failure is reported
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:
            rc = max(abs(p.returncode) for p in failed)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
        rc = max(rc, abs(p.returncode or 0))
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode("utf8", "replace"))
    sys.stdout.flush()
    return rc


def build_shard(args, rank, world):
    """This rank's piece of the node's batch: contiguous utterances with a near-equal share of the node's samples."""
    from nvspeechplayer_amd import workloads
    from nvspeechplayer_amd.sharding import shard_bounds
    per_gpu = args.utterances or workloads.PER_GPU[args.workload]
    node_utt = per_gpu * world
    counts = workloads.sample_counts(args.workload, node_utt)
    bounds = shard_bounds(counts, world)
    first, n = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
    batch = workloads.make(args.workload, n, first=first)
    assert batch.n_utt == n and int(batch.sample_counts().sum()) == int(counts[first:first + n].sum())
    return batch, {"node_utterances": node_utt, "node_samples": int(counts.sum()), "first_utterance": first, "utterances": n,
                   "bounds": [int(b) for b in bounds]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # first: its HIP runtime is then the one the engine library binds to
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer, workloads
    from nvspeechplayer_amd.sharding import reduce_throughput

    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path); --dry-run exercises the host side only")
    ndev = 0 if dry else torch.cuda.device_count()
    device = 0 if dry else local_rank % ndev
    if not dry:
        torch.cuda.set_device(device)
    dist = None
    shared_device = dry or world > ndev   # self-tests on a 1-GPU box: ranks share a device, RCCL cannot
    # BENCH_FORCE_PG=1: the process group of the N > 1 run -- RCCL ("nccl") with device_id, the barrier, both reductions on
    # device tensors -- at world size 1, so that the branch can be exercised on a one-GPU box (tests/test_gpu_parity.py)
    force_pg = world == 1 and not dry and os.environ.get("BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        import torch.distributed as dist
        if force_pg:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1,
                                    device_id=torch.device("cuda", device))
        elif shared_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        assert dist.get_world_size() == world and dist.get_rank() == rank

    if world > 1 and "SPEECHPLAYER_PLAN_THREADS" not in os.environ:
        # the ranks of a node share its host cores (a 16-CPU quota for 8 ranks on the driver's box): setUtterances plans with up to 8
        # threads by default, which at N = 8 would be 64 threads on 16 cores -- outside the timed region, but inside the driver's timeout
        os.environ["SPEECHPLAYER_PLAN_THREADS"] = str(max(1, usable_cores() // world))
    t_build = time.perf_counter()
    batch, shard = build_shard(args, rank, world)
    t_build = time.perf_counter() - t_build
    samples = int(batch.sample_counts().sum())
    bp = None
    t_set = t_set_again = 0.0
    t_set_pinned = None
    compact = None
    if not dry:
        bp = BatchPlayer(batch["sr"], device=device, mode=args.mode, layout=args.layout)
        t_set = time.perf_counter()
        bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"],
                         batch["isnull"], batch["seeds"])
        t_set = time.perf_counter() - t_set       # classification, lane packing, track planning, uploads: outside the timed region
        assert bp.totalSamples == samples
        # the same call again: what a batch costs a player that has set one before (device buffers allocated, the host's per-frame
        # scratch sized and touched) -- the first call above also pays the allocations
        t_set_again = time.perf_counter()
        bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"],
                         batch["isnull"], batch["seeds"])
        t_set_again = time.perf_counter() - t_set_again
        t_set_pinned = None
        if world == 1 and not args.no_extras:
            # and with the frames in page-locked memory the library handed out (speechPlayer_hostAlloc): one DMA, classified and hashed on the device
            from nvspeechplayer_amd import host_array
            try:
                fr = host_array(batch["frames"].shape, np.float64)
            except MemoryError:
                fr = None      # (a box that cannot page-lock the frames: the figure stays null)
            if fr is not None:
                fr[...] = batch["frames"]
                for _ in range(2):
                    t_set_pinned = time.perf_counter()
                    bp.setUtterances(batch["frame_start"], fr, batch["min"], batch["fade"], batch["index"], batch["isnull"], batch["seeds"])
                    t_set_pinned = time.perf_counter() - t_set_pinned
                del fr
        if world == 1 and not args.no_extras and args.workload in ("cfg2", "cfg4"):
            # the same batch handed over as IPA text (speechPlayer_batch_setIpa / _setIpaVoices): the producer builds the distinct frame lists
            # as 32-byte records, the device expands them, utterances share their list's frames -- "build" is numpy writing the arguments
            try:
                spec_fn = workloads.cfg2_spec if args.workload == "cfg2" else workloads.cfg4_spec
                c_spec = c_set = c_again = 0.0
                cbp = BatchPlayer(batch["sr"], device=device, mode=args.mode, layout=args.layout)
                for rep in range(3):
                    t = time.perf_counter(); spec = spec_fn(shard["utterances"], first=shard["first_utterance"]); c_spec = time.perf_counter() - t
                    t = time.perf_counter(); cbp.setIpa(**spec); dt = time.perf_counter() - t
                    if rep == 0:
                        c_set = dt
                    c_again = dt
                assert cbp.totalSamples == samples
                cbp.synthesize(); bp.synthesize()
                c_equal = cbp.digest() == bp.digest()
                c_ms = float(np.mean(cbp.time(5)))
                compact = {"arguments_s": round(c_spec, 4), "set_ipa_first_call_s": round(c_set, 4), "set_ipa_s": round(c_again, 4),
                           "build_plus_set_s": round(c_spec + c_again, 4), "kernel_ms": c_ms, "pcm_equal_to_the_plain_batch": bool(c_equal),
                           "resident_frames": None, "what": "speechPlayer_batch_setIpa%s on the same utterances: producer + records + device-side frame expansion + planning on the distinct lists"
                                                           % ("" if args.workload == "cfg2" else "Voices (32 defined voices)")}
                cbp.close()
            except Exception as e:      # noqa: BLE001
                compact = {"error": "%s: %s" % (type(e).__name__, e)}

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    if args.warmup > 0 and not dry:
        bp.time(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = bp.time(args.steps) if not dry else np.zeros(args.steps)      # K launches, each between two HIP events on the launch stream
    barrier()
    elapsed = time.perf_counter() - t0
    red_dev = "cpu" if (shared_device and not force_pg) else "cuda"
    elapsed, total_samples = reduce_throughput(elapsed, samples, dist, device=red_dev)   # max / sum over ranks
    info = bp.kernelInfo() if not dry else {}
    alg_bytes = batch.algorithmic_bytes()

    # N > 1: BASELINE's node-wide configurations at their node sizes, as extra keys: configs[3] (world x 125 000 utterances cut to
    # 0.5 s: 10^6 at N = 8) and configs[4] (world x 32 voice variants x 16 384 utterances: 256 x 16 384 at N = 8).  Same deal (contiguous
    # shards of near-equal sample count), same barriers and reductions, 5 launches each.  --dry-run: the deal alone (closed-form lengths).
    node_extras = {}
    if world > 1 and not args.utterances and not args.no_extras:
        from nvspeechplayer_amd.sharding import shard_bounds
        if bp is not None:
            bp.close(); bp = None
        head_name, head_sr, head_frames = batch["name"], batch["sr"], int(len(batch["min"]))
        batch = None
        for key, wl, launches in (("cfg3_node", "cfg3", 5), ("cfg4_node", "cfg4", 5)):
            per_gpu = workloads.PER_GPU[wl]
            counts = workloads.sample_counts(wl, per_gpu * world)
            bounds = shard_bounds(counts, world)
            first, n = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
            mine = int(counts[first:first + n].sum())
            x_ms, x_el, x_info, x_build, x_set = [0.0], 0.0, {}, 0.0, 0.0
            if not dry:
                # the shard in compact form (tests/test_gpu_compact.py: same PCM as the plain batch): configs[3] as the 512 cut frame lists its
                # utterances share, configs[4] as IPA text with a defined voice per variant -- megabytes per rank instead of 0.6 / 4.7 GB of frames
                x = BatchPlayer(head_sr, device=device, mode=args.mode, layout=args.layout)
                x_build = time.perf_counter()
                if wl == "cfg3":
                    lists, list_of, seeds = workloads.shared("cfg3", n, first=first)
                    x_build = time.perf_counter() - x_build
                    x_set = time.perf_counter()
                    x.setUtterancesShared(lists["frame_start"], lists["frames"], lists["min"], lists["fade"], list_of, lists["index"], lists["isnull"], seeds)
                else:
                    spec = workloads.cfg4_spec(n, first=first)
                    x_build = time.perf_counter() - x_build
                    x_set = time.perf_counter()
                    x.setIpa(**spec)
                x_set = time.perf_counter() - x_set
                assert x.totalSamples == mine
                x.time(1)
                barrier()
                t0 = time.perf_counter()
                x_ms = x.time(launches)
                barrier()
                x_el = time.perf_counter() - t0
                x_info = x.kernelInfo()
                x.close()
            x_el, x_total = reduce_throughput(x_el, mine, dist, device=red_dev)
            node_extras[key] = {"workload": "BASELINE configs[%s] at its node size for %d GPUs" % (wl[-1], world), "node_utterances": per_gpu * world,
                                "node_samples": int(counts.sum()), "total_samples_all_ranks": x_total, "launches": launches,
                                "value": None if dry else x_total * launches / x_el, "unit": "samples/s",
                                "ms_per_launch": None if dry else x_el / launches * 1e3, "kernel_ms_rank0": None if dry else float(np.mean(x_ms)),
                                "shard_bounds": [int(v) for v in bounds],
                                "host_rank0": {"build_batch_s": round(x_build, 3), "set_utterances_s": round(x_set, 3),
                                               "plan_threads": os.environ.get("SPEECHPLAYER_PLAN_THREADS")},
                                "tracks_rank0": None if dry else {"tracked_utterances": x_info.get("tracked_utterances"), "tracks": x_info.get("tracks"), "mbytes": x_info.get("track_mbytes")}}
    else:
        head_name, head_sr, head_frames = batch["name"], batch["sr"], int(len(batch["min"]))

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        out = {
            "metric": "audio samples/sec (whole node) at 22.05 kHz Klatt synth, batch-N utterances",
            "value": None if dry else total_samples * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": head_name, "utterances_per_gpu": shard["utterances"], "samples_per_gpu": samples,
                       "frames_per_gpu": head_frames, "sample_rate": head_sr, "mode": args.mode, "layout": args.layout,
                       "coefficient_tracks": None if dry else {"tracked_utterances": info["tracked_utterances"], "tracks": info["tracks"], "mbytes": info["track_mbytes"]},
                       "node_utterances": shard["node_utterances"], "node_samples": shard["node_samples"],
                       "shard_bounds": shard["bounds"], "process_group": None if dist is None else dist.get_backend(),
                       "world_size": 1 if dist is None else dist.get_world_size(),
                       "rccl_ranks": dist.get_world_size() if (dist is not None and dist.get_backend() == "nccl") else None,
                       "host": {"build_batch_s": round(t_build, 3), "set_utterances_s": round(t_set, 3), "set_utterances_again_s": round(t_set_again, 3),
                                "set_utterances_page_locked_frames_s": None if t_set_pinned is None else round(t_set_pinned, 3),
                                "from_ipa": compact,
                                "note": "outside the timed region: the frame producer (build) and speechPlayer_batch_setUtterances (classification, "
                                        "lane packing, track planning, uploads); a batch is set once and synthesised many times.  from_ipa: the same "
                                        "utterances handed over as text, pitches and seeds (compact form)"},
                       "parallelism": "node batch cut into %d contiguous shards of near-equal sample count, one process per GPU, no collective on the data path" % world},
        }
        out.update(node_extras)
        if dry:
            out["dry_run"] = True
            out["total_samples_all_ranks"] = total_samples
        else:
            out["realtime_factor"] = total_samples * args.steps / elapsed / head_sr
            out["config"]["host"]["first_launch_end_to_end_samples_per_s"] = samples / (t_set + k_ms * 1e-3)
            if compact and "build_plus_set_s" in compact:
                out["config"]["host"]["from_ipa_launch_end_to_end_samples_per_s"] = samples / (compact["build_plus_set_s"] + k_ms * 1e-3)
            achieved = alg_bytes / (k_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": kernel_name(info), "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                    "wavefronts": info["wavefronts"], "vgprs": info["vgprs"], "lds_bytes": info["lds_bytes"], "scratch_bytes": info["scratch_bytes"],
                    "note": "declared bound: HBM (2 B per sample + 388 B per frame). What binds is f64 VALU issue with two barrier-coupled waves per SIMD: "
                            "see `valu` (ONE figure: issue cycles of the VALU instructions executed / issue cycles the 1024 SIMDs had at the held clock) "
                            "and DESIGN.md section 4, Roofline.  The headline leans on the benchmark's duplicates (8 sentences x 64 pitches share 105 tracks): "
                            "`headline_without_tracks` and `general_case` on this line are the same sample count without that help"}
            # HBM bytes and VALU instructions per launch come from the PMC passes of tools/profile.sh (bench.py cannot run rocprofv3 on
            # itself); the file names the engine sources it was collected from, and is ignored when they have changed since
            try:
                pj = json.load(open(PMC_FILE))
                ent = pj.get(args.workload)
                if ent and not args.utterances and world == 1 and args.mode == 0 and args.layout == -1:
                    if pj.get("engine_sources_sha") == engine_source_digest():
                        roof["traffic"] = ent["hbm_bytes_per_launch"]
                        roof["traffic_source"] = "profiles/r6_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, bytes per launch)"
                        insts = ent["valu_insts_per_launch"]
                        f64i = ent.get("valu_f64_insts_per_launch")
                        clock = ent.get("held_clock_hz") or CLOCK_HZ_HELD_DEFAULT
                        if f64i is None:
                            f64i = insts        # (no f64 pass in the file: every instruction billed as f64 -- an upper bound)
                        cycles = F64_ISSUE_CYCLES * f64i + OTHER_ISSUE_CYCLES * (insts - f64i)
                        avail = SIMDS * clock * (k_ms * 1e-3)
                        roof["valu"] = {"insts_per_launch": insts, "f64_insts_per_launch": f64i, "issue_cycles_per_launch": cycles,
                                        "held_clock_hz": clock, "frac": cycles / avail, "frac_if_every_instruction_were_f64": F64_ISSUE_CYCLES * insts / avail,
                                        "unit": "VALU issue cycles used / available", "insts_per_64_samples": insts * 64.0 / samples,
                                        "formula": "(4 x f64 wave-instructions + 2 x other VALU wave-instructions) / (1024 SIMDs x held clock x kernel_ms); "
                                                   "counts: SQ_INSTS_VALU and SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per launch (separate rocprofv3 --pmc passes), "
                                                   "clock: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration -- all from profiles/r6_pmc.json, collected on these kernel sources"}
                    else:
                        roof["traffic_source"] = "profiles/r6_pmc.json is stale (kernels changed since it was collected): traffic and valu omitted"
            except Exception:
                pass
            out["roofline"] = roof
            if world == 1 and args.mode == 0 and not args.no_extras:
                # same batch in MODE_FAST (fused multiply-adds; held to <= 1 LSB and <= 5 one-LSB differences per million samples against the
                # oracle, not bit-exact: the tracked stages measured 0 differences on the test corpus, the direct stages' pole recurrences 36 in
                # 8.5 M samples of 350 000-sample fades)
                bp.setOption("mode", 1)
                bp.time(1)
                fast_ms = float(np.mean(bp.time(max(20, args.steps))))
                bp.setOption("mode", 0)
                out["mode_fast"] = {"value": samples / (fast_ms * 1e-3), "unit": "samples/s", "kernel_ms": fast_ms,
                                    "roofline_frac": alg_bytes / (fast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if world == 1 and args.mode == 0 and not args.no_extras and info.get("tracked"):
                # the same batch without tracks (every stage interpolates and evaluates exp/cos itself, as in round 1), and with every
                # utterance's frame list rotated by a random amount -- same lengths, ~24 different timings per sentence in random
                # order: the lane packing puts equally timed utterances side by side again
                def timed(b, tracks, sort=1):
                    x = BatchPlayer(b["sr"], device=device, mode=args.mode, layout=args.layout)
                    x.setOption("tracks", tracks)
                    x.setOption("sort", sort)
                    x.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
                    x.time(1)
                    ms = float(np.mean(x.time(10)))
                    x.close()
                    return {"value": samples / (ms * 1e-3), "unit": "samples/s", "kernel_ms": ms, "roofline_frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                out["tracks_off"] = timed(batch, 0)
                rot = workloads.rotated(batch)
                out["rotated_frame_lists"] = dict(timed(rot, 1), tracks_off=timed(rot, 0))
                # and with every frame's duration scaled by its own random factor: no two utterances share a timing or a length, so
                # no wavefront can be packed with equally timed utterances -- a batch of unrelated sentences
                jit = workloads.jittered(batch)
                out["jittered_durations"] = dict(timed(jit, 1), tracks_off=timed(jit, 0))
                # and the benchmarked batch itself without the engine's sort by length and timing: wavefronts of 64 arbitrary neighbours
                out["unsorted"] = dict(timed(batch, 1, 0), tracks_off=timed(batch, 0, 0))
                # The general case: nothing shared AND nothing aligned (every frame's formants scaled by its own factor on top of the jitter:
                # 65 536 different sentences in different voices).  Its tracks would take 57 GB: the planner gives up, the direct stages
                # (klatt_direct.h) compute every fade sample's coefficients in place -- polynomials in MODE_EXACT, pole recurrences in MODE_FAST;
                # "legacy": the stages with the frame state machine, which ran such batches until round 4.
                def timed_mode(b, mode, direct, tracks=1):
                    x = BatchPlayer(b["sr"], device=device, mode=mode, layout=args.layout)
                    x.setOption("tracks", tracks)
                    x.setOption("direct", direct)
                    x.setUtterances(b["frame_start"], b["frames"], b["min"], b["fade"], b["index"], b["isnull"], b["seeds"])
                    x.time(1)
                    ms = float(np.median(x.time(6)))
                    xi = x.kernelInfo()
                    n = x.totalSamples
                    x.close()
                    return {"value": n / (ms * 1e-3), "unit": "samples/s", "kernel_ms": ms, "roofline_frac": b.algorithmic_bytes() / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "kernel": kernel_name(xi), "direct_utterances": xi.get("direct_utterances"), "tracked_utterances": xi.get("tracked_utterances"),
                            "scratch_bytes": xi.get("scratch_bytes")}
                alld = workloads.all_different(batch)
                out["all_different"] = dict(timed_mode(alld, 0, 1), mode_fast=timed_mode(alld, 1, 1), legacy=timed_mode(alld, 0, 0))
                dst = workloads.distinct(batch)
                out["distinct"] = dict(timed_mode(dst, 0, 1), mode_fast=timed_mode(dst, 1, 1), direct_forced=timed_mode(dst, 0, 2), direct_forced_fast=timed_mode(dst, 1, 2))
                out["jittered_durations"]["mode_fast"] = timed_mode(jit, 1, 1)
                out["jittered_durations"]["direct_fast"] = timed_mode(jit, 1, 2, tracks=0)
                # the three numbers that belong together: the headline, the same batch without its tracks, the general case
                out["headline_without_tracks"] = dict(out["tracks_off"], what="the benchmarked batch with `tracks` off: every stage interpolates and evaluates exp / cos itself")
                out["general_case"] = dict({k: v for k, v in out["all_different"].items() if k != "legacy"},
                                           what="nothing shared, nothing aligned (workloads.all_different: every frame's duration, fade and formants scaled by factors of its own): "
                                                "the direct stages, two workgroups per CU; mode_fast: pole recurrences instead of polynomials")
            if world == 1 and not args.utterances and not args.no_extras:
                # the other single-GPU configuration (BASELINE configs[1], 4096 steady vowels) and its recipe at 65 536 utterances
                bp.close()
                for key, wl, n, reps in (("cfg1", "cfg1", 4096, 400), ("cfg1_recipe_at_batch_65536", "cfg1", 65536, 60)):
                    if key == "cfg1" and args.workload == "cfg1":
                        continue
                    xb = workloads.make(wl, n)
                    bp = BatchPlayer(xb["sr"], device=device, mode=args.mode, layout=args.layout)
                    bp.setUtterances(xb["frame_start"], xb["frames"], xb["min"], xb["fade"], xb["index"], xb["isnull"], xb["seeds"])
                    bp.time(2)
                    x_ms = float(np.mean(bp.time(reps)))
                    xi = bp.kernelInfo()
                    out[key] = {"workload": xb["name"], "value": bp.totalSamples / (x_ms * 1e-3), "unit": "samples/s", "kernel_ms": x_ms,
                                "roofline_frac": xb.algorithmic_bytes() / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": kernel_name(xi)}
                    bp.close()
                bp = None
            if world == 1 and not args.utterances and not args.no_extras and args.workload == "cfg2":
                # what the timed region leaves out, taken in: sustained throughput over eight distinct batches, host work overlapped
                batch_keep = batch
                # (sixteen batches: the first one's setUtterances has nothing to hide behind, and eight batches made that ramp a fifth of the figure)
                def extra(**kw):
                    # an extra must not cost the line its headline: a box that cannot page-lock 10 GB (or runs out of anything else) gets the reason instead
                    try:
                        return pipeline_extra(device, args.mode, args.layout, **kw)
                    except Exception as e:      # noqa: BLE001
                        return {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
                same = dict(n_batches=16, players=6, workers=4)      # every variant with the same counts
                out["pipeline_from_ipa"] = extra(source="ipa", **same)
                out["pipeline_from_ipa"]["with_pcm_to_host"] = extra(source="ipa", copy_out=True, n_batches=8, players=6, workers=4)
                out["pipeline"] = extra(**same)
                out["pipeline"]["with_pcm_to_host"] = extra(copy_out=True, n_batches=8, players=6, workers=4)
                out["pipeline"]["pageable_host_buffers"] = dict(extra(pinned=False, **same), with_pcm_to_host=extra(n_batches=8, players=6, workers=4, copy_out=True, pinned=False))
                batch = batch_keep
            if world == 1 and not args.no_extras:
                for key, fn in (("single_stream", single_stream_extra), ("live_handles", live_handles_extra), ("cfg0_cpu", cfg0_cpu_extra)):
                    try:
                        out[key] = fn()
                    except Exception as e:      # noqa: BLE001
                        out[key] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(batch, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if bp is not None:
        bp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
