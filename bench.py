#!/usr/bin/env python3
"""bench.py -- throughput of the Klatt hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg1|cfg2|cfg3] [--mode 0]

A step is one pass of the synthesis kernel over one batch of synthetic frame streams
that is already resident in HBM (frames in, int16 PCM out, both in HBM).  At N = 1 the
workload is BASELINE.json configs[1] (4096 steady vowels x 1 s); with N > 1 every rank
(one process per GPU, launched by torch.distributed.run) synthesises its own batch of
the same shape with different utterance numbers (weak scaling, no collective on the
data path -- utterances are independent).  Rank 0 prints ONE JSON line.

The K timed launches are individually bracketed by HIP events on the engine's stream
(speechPlayer_batch_time); their mean is the kernel duration the roofline uses, the
wall clock around all K (barrier + device synchronize on both sides, max over ranks)
gives `value`.  The CPU baseline is the oracle (oracle/klatt_oracle.c, "port") timed
on this box's host cores on the same workload, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--utterances", type=int, default=0, help="override the batch size per GPU")
    ap.add_argument("--mode", type=int, default=0, help="arithmetic mode (include/speechPlayer_batch.h)")
    ap.add_argument("--layout", type=int, default=-1, help="-1: engine's choice, 1: stage-parallel workgroups, 0: one wavefront per 64 utterances")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the extra measurement of the same recipe at 65536 utterances")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall time to spend on the CPU baseline")
    return ap.parse_args()


def kernel_name(info):
    """The kernel of the batch's largest group (speechPlayer_batch_kernelInfo)."""
    if info.get("lane_pipelined"):
        return "klatt_lanepipe (cascade across lanes, %d-sample hand-overs)" % info["stage_parallel_chunk"]
    if info["stage_parallel_chunk"]:
        return "klatt_systolic (stage-parallel%s, %d-sample hand-overs)" % (", nasal-free" if info.get("nasal_free") else "", info["stage_parallel_chunk"])
    return "klatt_synthesize (lane kernel)"


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
    """Oracle on the host cores: one OpenMP thread per core, each synthesising whole utterances of the
    same workload (no shared state: every utterance has its own player and noise stream).  The
    workload is repeated until about `target_seconds` of wall time have been spent."""
    from tests import oracle
    cores = usable_cores()
    probe = batch.slice(0, min(8, batch.n_utt))
    t0 = time.perf_counter()
    _, _, total = oracle.batch_synthesize(batch["sr"], probe, threads=1)
    one_core = total / (time.perf_counter() - t0)
    oracle.batch_synthesize(batch["sr"], batch.slice(0, min(batch.n_utt, cores)), threads=cores)   # start the thread pool
    done, reps, t0 = 0, 0, time.perf_counter()
    while True:
        _, _, total = oracle.batch_synthesize(batch["sr"], batch, threads=cores)
        done += total
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or reps >= 200:
            break
    return {"value": done / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "the whole workload (%d utterances) x %d passes = %d samples on %d OpenMP threads in %.1f s; "
                      "1 thread: %.3g samples/s" % (batch.n_utt, reps, done, cores, dt, one_core)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # first: its HIP runtime is then the one the engine library binds to
    import numpy as np
    from nvspeechplayer_amd import BatchPlayer, workloads
    from nvspeechplayer_amd.sharding import reduce_throughput

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    ndev = torch.cuda.device_count()
    device = local_rank % ndev
    torch.cuda.set_device(device)
    dist = None
    shared_device = world > ndev          # only in self-tests on a 1-GPU box: ranks share a device, RCCL cannot
    if world > 1:
        import torch.distributed as dist
        if shared_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    n_utt = args.utterances or {"cfg1": 4096, "cfg2": 65536, "cfg3": 131072, "cfg4": 32768}[args.workload]
    batch = workloads.make(args.workload, n_utt, first=rank * n_utt)
    bp = BatchPlayer(batch["sr"], device=device, mode=args.mode, layout=args.layout)
    bp.setUtterances(batch["frame_start"], batch["frames"], batch["min"], batch["fade"], batch["index"],
                     batch["isnull"], batch["seeds"])
    samples = bp.totalSamples
    assert samples == int(batch.sample_counts().sum())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        bp.time(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = bp.time(args.steps)          # K launches, each between two HIP events on the launch stream
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_samples = reduce_throughput(elapsed, samples, dist, device="cpu" if shared_device else "cuda")   # max / sum over ranks

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = batch.algorithmic_bytes()
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        info = bp.kernelInfo()
        traffic = None   # HBM bytes per launch from the PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on itself)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if args.workload in tj and not args.utterances:
                traffic = tj[args.workload]["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "audio samples/sec (whole node) at 22.05 kHz Klatt synth, batch-N utterances",
            "value": total_samples * args.steps / elapsed,
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
            "config": {"workload": batch["name"], "utterances_per_gpu": n_utt, "samples_per_gpu": int(samples),
                       "frames_per_gpu": int(bp.totalFrames), "sample_rate": batch["sr"], "mode": args.mode, "layout": args.layout,
                       "parallelism": "utterances sharded over %d GPU(s), no collective" % world},
            "realtime_factor": total_samples * args.steps / elapsed / batch["sr"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r1_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if traffic else None,
                         "kernel": kernel_name(info), "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "wavefronts": info["wavefronts"], "vgprs": info["vgprs"], "lds_bytes": info["lds_bytes"],
                         "note": "f64 VALU issue binds before HBM (a sample is a strict recurrence: its time per sample, not its bytes, "
                                 "bounds a launch); see DESIGN.md section 4"},
        }
        if world == 1 and args.mode == 0 and not args.no_large_batch:
            # same batch in MODE_FAST (fused multiply-adds; identical PCM on every test so far, not guaranteed bit-exact)
            bp.setOption("mode", 1)
            bp.time(1)
            fast_ms = float(np.mean(bp.time(max(3, args.steps // 2))))
            bp.setOption("mode", 0)
            out["mode_fast"] = {"value": samples / (fast_ms * 1e-3), "unit": "samples/s", "kernel_ms": fast_ms,
                                "roofline_frac": alg_bytes / (fast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.workload == "cfg1" and not args.utterances and not args.no_large_batch:
            # The metric is "batch-N utterances": the same recipe at N = 65 536 fills the chip (2 workgroups per CU).
            bp.close()
            big = workloads.make("cfg1", 65536)
            bp = BatchPlayer(big["sr"], device=device, mode=args.mode, layout=args.layout)
            bp.setUtterances(big["frame_start"], big["frames"], big["min"], big["fade"], big["index"], big["isnull"], big["seeds"])
            bp.time(1)
            big_ms = float(np.mean(bp.time(3)))
            big_bytes = big.algorithmic_bytes()
            out["same_recipe_at_batch_65536"] = {
                "value": bp.totalSamples / (big_ms * 1e-3), "unit": "samples/s", "kernel_ms": big_ms,
                "roofline_frac": big_bytes / (big_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "not the headline config; shows the occupancy limit of a 4096-utterance batch"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    bp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
