#!/usr/bin/env python3
"""bench.py — headline benchmark of the failure-aware classification path.

Workload (BASELINE.json `metric`, configs[2]): ResNet-50, MC-Dropout T=30 (dropout
after every residual block, p=0.1 — the `all_blocks` policy, 225.1 GFLOP per
frame algorithmic with the deterministic prefix computed once), 224x224 frames
with ImageNet-C style Gaussian noise severity 3, batch 256 per GPU.  A "step" is
one classify() of one batch: frames resident in HBM -> (label, confidence) per
frame.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every
rank classifies its own 256-frame shard of a 256*N global batch (weak scaling) and
the packed (label, confidence) records are all-gathered.

`python bench.py --gpus N` with no torchrun environment starts the N ranks itself
(a child `python -m torch.distributed.run ... bench.py`), BEFORE this process has
imported torch or touched the GPU, and exits with the child's return code.

Prints ONE JSON line on rank 0 (see README / DESIGN.md §6 for the fields).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RESNET50_GMAC = 4.089  # per frame, single pass
VIT_B16_GMAC = 17.56
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
KERNEL_SOURCES = ("failure_aware_vision_amd/csrc/fav_kernels.hpp", "failure_aware_vision_amd/csrc/fav.hip")


def algorithmic_gflop_per_frame(policy: str, T: int) -> float:
    prefix = {"none": RESNET50_GMAC, "all_blocks": 0.349, "layer4+fc": 3.278, "last_layer": 4.087}[policy]
    if policy == "none":
        return 2 * RESNET50_GMAC
    return 2 * (prefix + T * (RESNET50_GMAC - prefix))


def kernel_source_sha() -> str:
    """SHA-256 over the kernel sources a PMC profile was measured on (profiles/pmc_traffic.json records it)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic():
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes of the same command
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; separate
    passes; tools/profile.sh -> profiles/pmc_traffic.json).  The file records the SHA-256 of the
    kernel sources it was measured on; when that no longer matches the sources in this tree the
    figure is stale and None is reported (the roofline line then says `traffic: null`)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
        if d.get("kernel_source_sha256") != kernel_source_sha():
            return {"stale": True, "measured_on": d.get("kernel_source_sha256"), "source": d.get("source")}
        return {"bytes_per_launch": d["conv_bytes_per_launch"], "fetch_bytes_per_launch": d["conv_fetch_bytes_per_launch"],
                "write_bytes_per_launch": d["conv_write_bytes_per_launch"], "source": d["source"],
                "kernel_source_sha256": d["kernel_source_sha256"]}
    except Exception:
        return None


def effective_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of the host, and oversubscribing it with one thread per host core is several times slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(blob, args, T, policy):
    """The oracle's torch-CPU port (fp32 MKL-DNN, same weights, corruption, masks and prefix caching) on a
    bounded sample of the same workload, timed on this host's cores: `cpu_frames` frames x T samples, the T
    suffix passes stacked into one batch (what a many-core host runs best), one warm-up call - which also
    generates the Philox masks, so they are inputs, outside the timed region - then `cpu_repeats` timed calls."""
    import torch
    from failure_aware_vision_amd import synth, weights
    from oracle import fav_oracle as O
    from oracle import torch_cpu as TC
    n = args.cpu_frames
    threads = args.cpu_threads if args.cpu_threads > 0 else min(torch.get_num_threads(), effective_cpus())
    torch.set_num_threads(threads)
    frames = synth.gaussian_noise_f32(synth.synthetic_frames_u8(n, 224, 224, seed=21), 3, seed=3)
    model = O.parse_blob(blob)
    net = TC.TorchNet(model)
    net.mask_cache = {}
    cfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(1, policy), p=args.dropout_p, seed=4)
    t0 = time.perf_counter()
    TC.classify(model, frames, cfg, net=net, stack_samples=True)          # warm-up + mask generation
    warm = time.perf_counter() - t0
    times = []
    for _ in range(max(1, args.cpu_repeats)):
        t0 = time.perf_counter()
        TC.classify(model, frames, cfg, net=net, stack_samples=True)
        times.append(time.perf_counter() - t0)
        if sum(times) > 45.0:                                               # keep the default run within minutes
            break
    dt = statistics.median(times)
    gf = algorithmic_gflop_per_frame(policy, T)
    return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "batch": n, "repeats": len(times), "seconds": times, "warmup_seconds": warm,
            "gflops": n * gf / dt, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "usable_cpus": effective_cpus(),
            "sample": f"{n} frames x T={T} ({policy}) = {n * T} suffix passes in one stacked batch, oracle/torch_cpu.py "
                      f"fp32 MKL-DNN on {torch.get_num_threads()} threads; 1 warm-up call (also generates the Philox "
                      f"masks: inputs, not timed) + {len(times)} timed calls, median {dt:.2f} s"}


def secondary_configs(blob, args, torch, synth, weights, Backend):
    """Cheap driver-timed lines for the other BASELINE configs (1 GPU, after the headline's timed region):
    configs[1] single pass, configs[3]'s per-GPU share (5 members, 32 frames), configs[4]'s per-GPU share
    (ViT-B/16, 64 frames).  fps = frames / wall time of `steps` calls bracketed by synchronize()."""
    out = {}

    def timed(be, frames, steps, warm=2):
        for _ in range(warm):
            be.classify(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            be.classify(frames)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    u8 = synth.synthetic_frames_u8(256, 224, 224, seed=21)
    frames = torch.from_numpy(synth.gaussian_noise_f32(u8, 3, seed=3)).cuda()
    try:
        be = Backend("resnet50", blob, max_batch=256)
        dt = timed(be, frames, 10)
        be.close()
        out["single_pass"] = {"config": "BASELINE configs[1]: ResNet-50 224x224 batch 256, single pass", "frames_per_s": 256 / dt,
                              "ms_per_call": dt * 1e3, "tflops": 256 * 2 * RESNET50_GMAC / dt / 1e3}
    except Exception as e:
        out["single_pass"] = {"error": str(e)}
    try:
        members = [blob] + [weights.make_synthetic("resnet50", seed=s)[0] for s in (2, 3, 4, 5)]
        be = Backend("resnet50", members, max_batch=32)
        dt = timed(be, frames[:32], 10)
        be.close()
        out["ensemble5@32"] = {"config": "BASELINE configs[3] per-GPU share: 5 x ResNet-50 members, 32 frames per call",
                               "frames_per_s": 32 / dt, "ms_per_call": dt * 1e3, "tflops": 32 * 5 * 2 * RESNET50_GMAC / dt / 1e3}
    except Exception as e:
        out["ensemble5@32"] = {"error": str(e)}
    try:
        vblob, _ = weights.make_synthetic_vit("vit_b16", seed=1)
        be = Backend("vit_b16", vblob, max_batch=64, temperature=1.5, conf_kind="entropy")
        dt = timed(be, frames[:64], 10)
        be.close()
        out["vit_b16@64"] = {"config": "BASELINE configs[4] per-GPU share: ViT-B/16, 64 frames per call, entropy confidence",
                             "frames_per_s": 64 / dt, "ms_per_call": dt * 1e3, "tflops": 64 * 2 * VIT_B16_GMAC / dt / 1e3}
    except Exception as e:
        out["vit_b16@64"] = {"error": str(e)}
    return out


def launch_command(gpus: int, argv, port: int | None = None):
    """The child command `python bench.py --gpus N` runs when no torchrun environment is present."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--samples", type=int, default=30, help="MC-Dropout T")
    ap.add_argument("--policy", default="all_blocks", choices=["none", "last_layer", "layer4+fc", "all_blocks"])
    ap.add_argument("--dropout-p", type=float, default=0.1)
    ap.add_argument("--chunk-a", type=int, default=0)
    ap.add_argument("--chunk-b", type=int, default=0)
    ap.add_argument("--regroup-block", type=int, default=-1)
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed repeats of the CPU baseline")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = the CPUs this job may use)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs (single pass, ensemble, ViT)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    # ---- rank layout; self-launch BEFORE anything initialises the GPU -------------------------
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
            sys.exit(subprocess.call(launch_command(args.gpus, sys.argv[1:]), env=env))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from failure_aware_vision_amd import Backend, classify_sharded, synth, weights

    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    T = args.samples if args.policy != "none" else 1
    blob, info = weights.make_synthetic("resnet50", seed=1)
    be = Backend("resnet50", blob, device=local_rank, max_batch=args.batch, n_samples=T, dropout_policy=args.policy,
                 dropout_p=args.dropout_p if args.policy != "none" else 0.0, seed=4,
                 chunk_a=args.chunk_a, chunk_b=args.chunk_b, regroup_block=args.regroup_block)
    n_total = args.batch * world
    start = rank * args.batch
    frames_u8 = synth.synthetic_frames_u8(args.batch, 224, 224, seed=21, start_id=start)
    frames = torch.from_numpy(synth.gaussian_noise_f32(frames_u8, 3, seed=3, start_id=start)).cuda()

    def step():
        return classify_sharded(be.classify, frames, n_total, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    profile = not args.no_profile
    if profile:
        be.set_profiling(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        labels, conf = step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    lat_ms = [a.elapsed_time(b) for a, b in ev]
    prof = be.get_profile() if profile else None
    be.set_profiling(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        fps = n_total * args.steps / elapsed
        gf = algorithmic_gflop_per_frame(args.policy, T)
        out = {
            "metric": "frames/sec, ResNet-50 MC-Dropout T=%d 224x224 batch=%d" % (T, args.batch),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "p50_latency_ms": statistics.median(lat_ms),
            "config": {"workload": "BASELINE configs[2]: ResNet-50 v1.5 + MC-Dropout T=%d (%s, p=%.2f), 224x224x3 fp32 "
                                   "frames with Gaussian noise severity 3, batch %d per GPU, synthetic frames and "
                                   "seeded synthetic weights" % (T, args.policy, args.dropout_p, args.batch),
                       "frames_per_gpu": args.batch, "global_batch": n_total, "mc_samples": T,
                       "dropout_policy": args.policy, "algorithmic_gflop_per_frame": gf,
                       "inputs": "resident in HBM before the timed region (H2D excluded)",
                       "parallelism": "batch sharded, 1 process per GPU, all-gather of (label, confidence)"},
            "achieved_tflops_algorithmic": fps * gf / 1000.0 / world,
            "labels_distinct": int(len(set(labels.cpu().tolist()))),
        }
        if prof is not None:
            cv = prof["conv_igemm"]
            secs = cv["ms"] * 1e-3
            total_ms = sum(v["ms"] for v in prof.values())
            tf = cv["flops"] / secs / 1e12 if secs > 0 else 0.0
            gbs = cv["bytes"] / secs / 1e9 if secs > 0 else 0.0
            common = {"kernel": "fav::conv_igemm_kernel + fav::conv3x3_halo_kernel + fav::bottleneck_tail_kernel + fav::entry_reduce_kernel "
                                "(every conv / fc launch of the timed steps)",
                      "launches": cv["launches"], "avg_launch_us": 1e3 * cv["ms"] / max(1, cv["launches"]),
                      "share_of_kernel_time": cv["ms"] / total_ms if total_ms > 0 else None,
                      "timing": "HIP events recorded on the launch stream around every launch, inside the timed region"}
            # Layer by layer the workload is HBM-bound overall (algorithmic FLOP/B below the machine
            # balance of 312 FLOP/B), so the binding roofline is HBM.
            tr = pmc_traffic()
            fresh = tr is not None and not tr.get("stale")
            out["roofline"] = dict(common, bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                                   frac=gbs / PEAK_HBM_GBS,
                                   traffic=tr["bytes_per_launch"] if fresh else None,   # HBM bytes per launch from the PMC passes
                                   traffic_detail=tr,
                                   algorithmic_bytes_per_launch=cv["bytes"] / max(1, cv["launches"]))
            out["roofline_mfma"] = dict(common, bound="mfma", achieved=tf, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                                        frac=tf / PEAK_BF16_TFLOPS, traffic=None)
            out["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in prof.items()}
    be.close()
    if rank == 0:
        if world == 1 and not args.no_extra:
            try:
                out["extra"] = secondary_configs(blob, args, torch, synth, weights, Backend)
            except Exception as e:   # secondary lines are reported, never required
                out["extra"] = {"error": str(e)}
        if args.cpu_frames > 0 and world == 1:   # the CPU baseline is timed on rank 0 of the 1-GPU run only
            try:
                out["cpu_baseline"] = cpu_baseline(blob, args, T, args.policy)
                out["gpu_over_cpu"] = fps / world / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
