#!/usr/bin/env python3
"""bench.py — benchmark of the failure-aware classification path.

Default workload (BASELINE.json `metric`, configs[2]): ResNet-50, MC-Dropout T=30 (dropout
after every residual block, p=0.1 — the `all_blocks` policy, 225.1 GFLOP per
frame algorithmic with the deterministic prefix computed once), 224x224 frames
with ImageNet-C style Gaussian noise severity 3, batch 256 per GPU.  A "step" is
one classify() of one batch: frames resident in HBM -> (label, confidence) per
frame.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every
rank classifies its own shard, the confidence head writes its packed (label, confidence)
records straight into the all-gather send slot and one all-gather moves them.

`--config` selects another BASELINE.json config as the measured line (same JSON contract,
same sharding + all-gather, its own `roofline`):
  mc30    configs[2]  ResNet-50 MC-Dropout T=30, 256 frames per GPU      (weak scaling; the default / headline)
  single  configs[1]  ResNet-50 single pass, 256 frames per GPU           (weak scaling)
  ens5    configs[3]  5-member ResNet-50 ensemble, global batch 256 / N   (strong scaling)
  vit     configs[4]  ViT-B/16 + temperature entropy, global batch 512 / N (strong scaling)
Without `--config` the line is the headline's, with the other configs' per-GPU shares under `extra`.

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
CONV_KERNELS = ("fav::conv_igemm_kernel + fav::conv3x3_halo_kernel + fav::bottleneck_tail_kernel + fav::entry_reduce_kernel + fav::stem7_pool_kernel "
                "(every conv / fc launch of the timed steps)")


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


def _cpu_threads(args, torch):
    threads = args.cpu_threads if args.cpu_threads > 0 else min(torch.get_num_threads(), effective_cpus())
    torch.set_num_threads(threads)
    return torch.get_num_threads()


def cpu_baseline_fp32_module(args, T, policy):
    """Second CPU figure: the headline algorithm as a PyTorch-CPU user would run it (oracle/torch_fp32.py): the synthetic checkpoint as a
    plain fp32 nn.Module (no bf16 anywhere, the library's fp32 MKL-DNN convolutions, conv + BatchNorm fused for inference
    by torch.nn.utils.fusion as any serving setup does), the deterministic prefix once, the T dropout samples of the
    suffix stacked (a few hundred virtual frames per pass) with in-place F.dropout masks from torch's own generator,
    mean of softmax.  Bounded sample: `cpu_frames` frames (b = 32), one
    warm-up call on a quarter of them, then two timed calls (a second figure beside the port's three; 120 s cap)."""
    import numpy as np
    import torch
    from failure_aware_vision_amd import synth, weights
    from oracle import torch_fp32 as TF
    n = args.cpu_frames
    threads = _cpu_threads(args, torch)
    net, meta = TF.load_synthetic("resnet50", seed=1)
    net = TF.fuse_for_inference(net)                              # conv + eval BatchNorm -> one fp32 conv, as for any serving
    x = synth.gaussian_noise_f32(synth.synthetic_frames_u8(n, 224, 224, seed=21), 3, seed=3)
    xn = (x - np.asarray(meta["mean"], np.float32)) / np.asarray(meta["std"], np.float32)
    xt = torch.from_numpy(np.ascontiguousarray(xn.transpose(0, 3, 1, 2))).contiguous(memory_format=torch.channels_last)
    sm = weights.site_mask_for(1, policy) if policy != "none" else 0
    gen_p = round(args.dropout_p * 256) / 256.0                   # the GPU path draws 8-bit thresholds
    t0 = time.perf_counter()
    TF.mc_dropout_probs(net, xt[:max(1, n // 4)], T, sm, gen_p, chunk=args.cpu_chunk)
    warm = time.perf_counter() - t0
    times = []
    while len(times) < max(1, min(2, args.cpu_repeats)) and (not times or sum(times) < 120.0):
        t0 = time.perf_counter()
        TF.mc_dropout_probs(net, xt, T, sm, gen_p, chunk=args.cpu_chunk)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    gf = algorithmic_gflop_per_frame(policy, T)
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "fp32_module",
            "batch": n, "repeats": len(times), "seconds": times, "warmup_seconds": warm,
            "gflops": n * gf / dt, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "usable_cpus": effective_cpus(),
            "sample": f"{n} frames x T={T} ({policy}) = {n * T} suffix passes in one stacked batch through the fp32 nn.Module "
                      f"of oracle/torch_fp32.py (conv+BN fused for inference, in-place F.dropout masks, torch {torch.__version__} MKL-DNN) on "
                      f"{threads} threads; 1 warm-up call on {max(1, n // 4)} frames + {len(times)} timed calls, median {dt:.2f} s"}


def cpu_baseline(blob, args, T, policy):
    """The oracle's torch-CPU port (oracle/torch_cpu.py) - the fastest CPU implementation of this workload here (2x the plain
    nn.Module below on the same cores): fp32 MKL-DNN convolutions with the bias fused, in-place residual / ReLU / mask
    multiply, the SAME weights, corruption, Philox masks and prefix caching as the GPU path, bf16 rounding at the layer
    boundaries as the numerical contract says.  `cpu_port_frames` frames x T samples, the suffix passes stacked into one
    batch; one warm-up call - which also generates the Philox masks, so they are inputs and not timed - then
    `cpu_repeats` timed calls (b = 32, three repeats: SURVEY.md section 8d; a box that needs more than 120 s for them stops early)."""
    import torch
    from failure_aware_vision_amd import synth, weights
    from oracle import fav_oracle as O
    from oracle import torch_cpu as TC
    n = args.cpu_port_frames
    threads = _cpu_threads(args, torch)
    frames = synth.gaussian_noise_f32(synth.synthetic_frames_u8(n, 224, 224, seed=21), 3, seed=3)
    model = O.parse_blob(blob)
    net = TC.TorchNet(model)
    net.mask_cache = {}
    cfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(1, policy), p=args.dropout_p, seed=4)
    t0 = time.perf_counter()
    TC.classify(model, frames, cfg, net=net, stack_samples=True)          # warm-up + mask generation
    warm = time.perf_counter() - t0
    times = []
    while len(times) < max(1, args.cpu_repeats) and (not times or sum(times) < 120.0):
        t0 = time.perf_counter()
        TC.classify(model, frames, cfg, net=net, stack_samples=True)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    gf = algorithmic_gflop_per_frame(policy, T)
    return {"value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "batch": n, "repeats": len(times), "seconds": times, "warmup_seconds": warm,
            "gflops": n * gf / dt, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
            "usable_cpus": effective_cpus(),
            "sample": f"{n} frames x T={T} ({policy}) = {n * T} suffix passes in one stacked batch, oracle/torch_cpu.py "
                      f"fp32 MKL-DNN on {threads} threads; 1 warm-up call (also generates the Philox "
                      f"masks: inputs, not timed) + {len(times)} timed calls, median {dt:.2f} s"}


# --------------------------------------------------------------------------------------------------------------
# The measured configurations
# --------------------------------------------------------------------------------------------------------------
def config_table(args):
    T = args.samples if args.policy != "none" else 1
    return {
        "mc30": dict(arch="resnet50", scaling="weak", per_gpu=args.batch, T=T, policy=args.policy,
                     metric="frames/sec, ResNet-50 MC-Dropout T=%d 224x224 batch=%d" % (T, args.batch),
                     workload="BASELINE configs[2]: ResNet-50 v1.5 + MC-Dropout T=%d (%s, p=%.2f), 224x224x3 fp32 "
                              "frames with Gaussian noise severity 3, batch %d per GPU, synthetic frames and "
                              "seeded synthetic weights" % (T, args.policy, args.dropout_p, args.batch),
                     gflop=algorithmic_gflop_per_frame(args.policy, T), bound="hbm"),
        "single": dict(arch="resnet50", scaling="weak", per_gpu=args.batch, T=1, policy="none",
                       metric="frames/sec, ResNet-50 single pass 224x224 batch=%d" % args.batch,
                       workload="BASELINE configs[1]: ResNet-50 v1.5 single deterministic pass, max-softmax confidence, 224x224x3 "
                                "fp32 frames with Gaussian noise severity 3, batch %d per GPU" % args.batch,
                       gflop=2 * RESNET50_GMAC, bound="hbm"),
        "ens5": dict(arch="resnet50", scaling="strong", global_batch=256, T=1, policy="none", members=5,
                     metric="frames/sec, 5-member ResNet-50 deep ensemble 224x224 global batch=256",
                     workload="BASELINE configs[3]: 5 independently seeded ResNet-50 members, mean of member softmax, global "
                              "batch 256 sharded over the GPUs (every rank runs all 5 members on its shard)",
                     gflop=5 * 2 * RESNET50_GMAC, bound="mfma"),
        "vit": dict(arch="vit_b16", scaling="strong", global_batch=512, T=1, policy="none",
                    metric="frames/sec, ViT-B/16 temperature-scaled entropy 224x224 global batch=512",
                    workload="BASELINE configs[4]: ViT-B/16 (MFMA attention path), confidence = 1 - H(softmax(z / 1.5)) / ln C, "
                             "global batch 512 sharded over the GPUs",
                    gflop=2 * VIT_B16_GMAC, bound="mfma"),
    }


def make_backend(name, c, args, n_max, device, weights, Backend, blob=None):
    if name == "vit":
        vblob, _ = weights.make_synthetic_vit("vit_b16", seed=1)
        return Backend("vit_b16", vblob, device=device, max_batch=n_max, temperature=1.5, conf_kind="entropy")
    if name == "ens5":
        members = [blob or weights.make_synthetic("resnet50", seed=1)[0]] + [weights.make_synthetic("resnet50", seed=s)[0] for s in (2, 3, 4, 5)]
        return Backend("resnet50", members, device=device, max_batch=n_max)
    if blob is None:
        blob = weights.make_synthetic("resnet50", seed=1)[0]
    if c["policy"] == "none":
        return Backend("resnet50", blob, device=device, max_batch=n_max)
    return Backend("resnet50", blob, device=device, max_batch=n_max, n_samples=c["T"], dropout_policy=c["policy"],
                   dropout_p=args.dropout_p, seed=4, chunk_a=args.chunk_a, chunk_b=args.chunk_b, regroup_block=args.regroup_block)


def make_frames(n, start, args, torch, synth, device_corrupt=False):
    """fp32 [0,1] frames with Gaussian noise severity 3, resident on the GPU.  Default: corrupted on the host (NumPy
    generator, the frames every fixture uses).  --device-corrupt: the clean uint8 frames are uploaded and corrupted by
    the on-device generator (corrupt.py: Philox noise keyed by the global frame index), before the timed region."""
    u8 = synth.synthetic_frames_u8(n, 224, 224, seed=21, start_id=start)
    if device_corrupt:
        from failure_aware_vision_amd.corrupt import Corruptor
        return Corruptor(seed=3).gaussian(torch.from_numpy(u8).cuda(), severity=3, first_index=start)
    return torch.from_numpy(synth.gaussian_noise_f32(u8, 3, seed=3, start_id=start)).cuda()


def rooflines(prof, bound, timing_note=None, wall_s=None):
    """`roofline` objects of the GEMM-shaped launches (conv / fc / ViT linear + attention: class conv_igemm) from the HIP
    events recorded on the launch streams around every launch.  `wall_s`: the configs whose launches overlap on several
    streams (ensemble members, the two halves of a ViT batch) - a launch's event duration then includes the time it
    shares the chip with the other streams' launches, so `achieved` is the class's algorithmic work over the wall time of
    the profiled region and the per-launch figure is kept as `achieved_per_launch_events`."""
    cv = prof["conv_igemm"]
    secs = cv["ms"] * 1e-3
    total_ms = sum(v["ms"] for v in prof.values())
    tf = cv["flops"] / secs / 1e12 if secs > 0 else 0.0
    gbs = cv["bytes"] / secs / 1e9 if secs > 0 else 0.0
    common = {"kernel": CONV_KERNELS, "launches": cv["launches"], "avg_launch_us": 1e3 * cv["ms"] / max(1, cv["launches"]),
              "share_of_kernel_time": cv["ms"] / total_ms if total_ms > 0 else None,
              "timing": timing_note or "HIP events recorded on the launch stream around every launch, inside the timed region"}
    tr = pmc_traffic()
    fresh = tr is not None and not tr.get("stale")
    hbm = dict(common, bound="hbm", achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS,
               traffic=tr["bytes_per_launch"] if fresh else None,   # HBM bytes per launch from the PMC passes (headline schedule)
               traffic_detail=tr, algorithmic_bytes_per_launch=cv["bytes"] / max(1, cv["launches"]))
    mfma = dict(common, bound="mfma", achieved=tf, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_BF16_TFLOPS, traffic=None,
                algorithmic_flop_per_launch=cv["flops"] / max(1, cv["launches"]))
    if wall_s:
        for r, work, scale in ((hbm, cv["bytes"], 1e9), (mfma, cv["flops"], 1e12)):
            r["achieved_per_launch_events"] = r["achieved"]
            r["achieved"] = work / wall_s / scale
            r["frac"] = r["achieved"] / r["peak"]
            r["traffic"] = None
    return (hbm, mfma) if bound == "hbm" else (mfma, hbm)


def measure(be, frames, n_total, rank, world, steps, warmup, torch, dist, classify_sharded, profile=True, separate=False):
    """The timed region: `steps` sharded classify calls between two fences (synchronize + barrier), max over ranks.
    profile: HIP events around every launch - inside the timed region (the headline: < 2 % of a 78 ms step), or with
    separate=True in a second pass of the same calls right after it (the 4-5 ms steps of the other configs, where
    ~200 event records per call would cost 10-20 % of the step).  -> (elapsed s, per-step ms, profile, labels, profiled wall s)"""
    def step():
        return classify_sharded(be, frames, n_total, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    if profile and not separate:
        be.set_profiling(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        labels, conf = step()
        b.record()
    fence()
    elapsed = time.perf_counter() - t0
    lat_ms = [a.elapsed_time(b) for a, b in ev]
    prof, prof_wall = None, elapsed
    if profile and separate:
        be.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        prof_wall = time.perf_counter() - t0
    if profile:
        prof = be.get_profile()
    be.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, lat_ms, prof, labels, prof_wall


SEPARATE_NOTE = ("HIP events recorded on the launch streams around every launch, in a second pass of the same calls right after the "
                 "timed region (recording ~200 events per call inside a 4-5 ms step would slow the step itself)")


def secondary_configs(blob, args, torch, dist, synth, weights, Backend, classify_sharded):
    """Driver-timed lines for the other BASELINE configs (1 GPU, after the headline's timed region), each with its own
    `roofline`: configs[1] single pass, configs[3]'s per-GPU share on an 8-GPU node (5 members, 32 frames) and
    configs[4]'s (ViT-B/16, 64 frames).  fps = frames / wall time of `steps` calls bracketed by synchronize()."""
    out = {}
    table = config_table(args)
    frames = make_frames(256, 0, args, torch, synth)
    for key, name, n in (("single_pass", "single", 256), ("ensemble5@32", "ens5", 32), ("vit_b16@64", "vit", 64)):
        c = table[name]
        try:
            be = make_backend(name, c, args, n, torch.cuda.current_device(), weights, Backend, blob)
            elapsed, lat, prof, _, pwall = measure(be, frames[:n], n, 0, 1, 10, 2, torch, dist, classify_sharded, separate=True)
            be.close()
            dt = elapsed / 10
            primary, other = rooflines(prof, c["bound"], SEPARATE_NOTE, pwall if name in ("ens5", "vit") else None)
            out[key] = {"config": c["workload"] + (" - per-GPU share of an 8-GPU node: %d frames per call" % n if c["scaling"] == "strong" else ""),
                        "frames_per_s": n / dt, "ms_per_call": dt * 1e3, "tflops": n * c["gflop"] / dt / 1e3,
                        "roofline": primary, "roofline_" + other["bound"]: other}
        except Exception as e:
            out[key] = {"error": str(e)}
    return out


def seam_latency(blob, torch, synth, Backend):
    """The reference's own operating point (main.py:122,160,205; video_source.py:29-30): ONE 320x240 uint8 frame per call
    through Backend.analyze_frame (upload, rule statistics kernel, classifier, one host sync, scalar scoring), 200 calls,
    for the headline estimator (T = 30) and a single pass, against the 33.3 ms tick of the 30 Hz loop."""
    out = {"budget_ms": 1000.0 / 30.0, "frame": "320x240x3 uint8 (host)", "calls": 200,
           "path": "Backend.analyze_frame: one upload, signal_stats_kernel + ResNet-50, one sync, dict as signal_analyzer.py:128-143"}
    frames = synth.synthetic_frames_u8(8, 240, 320, seed=5)
    for key, kw in (("mc30", dict(n_samples=30, dropout_policy="all_blocks", dropout_p=0.1, seed=4)), ("single_pass", {})):
        be = Backend("resnet50", blob, in_hw=(240, 320), max_batch=1, **kw)
        for i in range(10):
            be.analyze_frame(frames[i % 8])
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            r = be.analyze_frame(frames[i % 8])
            lat.append((time.perf_counter() - t0) * 1e3)
        be.close()
        lat.sort()
        out[key] = {"p50_ms": lat[100], "p95_ms": lat[190], "max_ms": lat[-1], "last_status": r["vision_status"]}
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
    ap.add_argument("--config", default=None, choices=["mc30", "single", "ens5", "vit"],
                    help="the BASELINE config to measure (default: the headline, configs[2], with the others under `extra`)")
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (mc30, single)")
    ap.add_argument("--samples", type=int, default=30, help="MC-Dropout T")
    ap.add_argument("--policy", default="all_blocks", choices=["none", "last_layer", "layer4+fc", "all_blocks"])
    ap.add_argument("--dropout-p", type=float, default=0.1)
    ap.add_argument("--chunk-a", type=int, default=0)
    ap.add_argument("--chunk-b", type=int, default=0)
    ap.add_argument("--regroup-block", type=int, default=-1)
    ap.add_argument("--device-corrupt", action="store_true", help="corrupt the frames with the on-device generator (corrupt.py)")
    ap.add_argument("--cpu-port-frames", type=int, default=32, help="frames in the CPU baseline sample, oracle/torch_cpu.py (SURVEY 8d: b = 32; 0 = skip)")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed repeats of the CPU baselines (SURVEY 8d: >= 3)")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the second CPU figure, the plain fp32 nn.Module (SURVEY 8d: b = 32; 0 = skip)")
    ap.add_argument("--cpu-chunk", type=int, default=240, help="virtual frames per stacked suffix pass of the fp32 module")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = the CPUs this job may use)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N > 1 path with all ranks on ONE GPU (records cross the host; not a measurement)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs and the seam latency")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    # dmabuf IPC: what RCCL needs on this host driver; set before anything loads the HIP runtime, also when the ranks
    # were started directly by torchrun
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ---- rank layout; self-launch BEFORE anything initialises the GPU -------------------------
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            env = dict(os.environ)
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
    from failure_aware_vision_amd import Backend, classify_sharded, shard_range, synth, weights

    rehearsal = world > 1 and args.dist_backend == "gloo"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())   # ranks share the GPUs there are
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    name = args.config or "mc30"
    c = config_table(args)[name]
    if c["scaling"] == "weak":
        n_total = c["per_gpu"] * world
    else:
        n_total = c["global_batch"]
    start, stop = shard_range(n_total, rank, world)
    n_local = stop - start
    blob, info = weights.make_synthetic("resnet50", seed=1)
    be = make_backend(name, c, args, max(1, -(-n_total // world)), local_rank, weights, Backend, blob)
    frames = make_frames(n_local, start, args, torch, synth, args.device_corrupt)
    T = c["T"]

    separate = name != "mc30"
    elapsed, lat_ms, prof, labels, pwall = measure(be, frames, n_total, rank, world, args.steps, args.warmup, torch, dist,
                                                   classify_sharded, profile=not args.no_profile, separate=separate)

    out = None
    if rank == 0:
        fps = n_total * args.steps / elapsed
        gf = c["gflop"]
        out = {
            "metric": c["metric"],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": c["scaling"],
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic" + (" (REHEARSAL: gloo, ranks share a GPU - not a measurement)" if rehearsal else ""),
            "p50_latency_ms": statistics.median(lat_ms),
            "config": {"workload": c["workload"],
                       "frames_per_gpu": n_local, "global_batch": n_total, "mc_samples": T,
                       "dropout_policy": c["policy"], "algorithmic_gflop_per_frame": gf,
                       "inputs": "resident in HBM before the timed region (H2D excluded)" +
                                 ("; corrupted by the on-device generator" if args.device_corrupt else ""),
                       "parallelism": "batch sharded, 1 process per GPU, all-gather of (label, confidence)"},
            # what the collective ran on: the backend torch.distributed reports and the rank count IT saw (N = 1: no process group)
            "dist": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "collective": "all_gather_into_tensor of 8-byte (label, confidence) records",
                      "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None)}
                     if world > 1 else {"backend": None, "world_size": 1}),
            "achieved_tflops_algorithmic": fps * gf / 1000.0 / world,
            "labels_distinct": int(len(set(labels.cpu().tolist()))),
        }
        if prof is not None:
            primary, other = rooflines(prof, c["bound"], SEPARATE_NOTE if separate else None, pwall if name in ("ens5", "vit") else None)
            out["roofline"] = primary
            out["roofline_" + other["bound"]] = other
            out["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in prof.items()}
    be.close()
    if rank == 0:
        if world == 1 and not args.no_extra and args.config is None:
            try:
                out["extra"] = secondary_configs(blob, args, torch, dist, synth, weights, Backend, classify_sharded)
            except Exception as e:   # secondary lines are reported, never required
                out["extra"] = {"error": str(e)}
            try:
                out["extra"]["seam_320x240"] = seam_latency(blob, torch, synth, Backend)
            except Exception as e:
                out["extra"]["seam_320x240"] = {"error": str(e)}
        if world == 1 and name == "mc30":   # the CPU baselines are timed on rank 0 of the 1-GPU headline run only
            if args.cpu_port_frames > 0:
                try:
                    out["cpu_baseline"] = cpu_baseline(blob, args, T, c["policy"])
                    out["gpu_over_cpu"] = fps / world / out["cpu_baseline"]["value"]
                except Exception as e:  # the baseline is reported, never required
                    out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            if args.cpu_frames > 0:
                try:
                    out["cpu_baseline_fp32_module"] = cpu_baseline_fp32_module(args, T, c["policy"])
                except Exception as e:
                    out["cpu_baseline_fp32_module"] = {"value": None, "kind": "fp32_module", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
