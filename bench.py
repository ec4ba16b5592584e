#!/usr/bin/env python3
"""Headline benchmark: PSMNet (stacked hourglass) forward, D=192, 384x1280, pairs/s.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU.  Under the driver's launcher
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
the ranks come from RANK / LOCAL_RANK / WORLD_SIZE.  A plain `python bench.py --gpus N` (no
WORLD_SIZE in the environment) starts the N rank processes itself -- child processes, started
before the parent has made any GPU call -- waits for them and exits non-zero if any failed.
Stereo pairs are independent, so the job shards them across ranks with no data-path
collective (weak scaling: one pair per GPU per step); RCCL carries only the timing barrier
and the max-over-ranks reduction.

A step = one forward pass over one synthetic KITTI-shaped pair already resident in HBM
(2-D towers -> cost volume -> 3-D trunk -> three soft-argmin heads, all three computed as
the reference does).  Weights: the reference's random initialisation, BN statistics and
head scale calibrated (dsmnet_amd/calibrate.py).  Compute dtype fp32, as the reference.

Rank 0 prints ONE JSON line: throughput (K steps between two barriers), per-step spread
(median / min from one HIP event per step), the per-kernel rooflines from a second,
instrumented pass (two HIP events per launch on the launch stream) and, at N=1, the CPU
baseline: the oracle restatement of the same forward on the same weights, per stage, timed on
the host cores.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "stereo pairs/sec forward, PSMNet D=192 384×1280, at 1/2/4/8 MI355X"
H, W, MAXDISP = 384, 1280, 192
# MI355X ceilings (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s spec, fp32-input
# MFMA 157.3 TFLOP/s dense (v_mfma_f32_32x32x2_f32; no xf32/TF32 on gfx950)
PEAK_HBM_GBS = 8000.0
HBM_COPY_CEILING_GBS = 6290.0      # measured float4 copy (same guide); reported beside the spec fraction
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak
# what a bare v_mfma_f32_16x16x32_f16 loop sustains on every SIMD of this chip with non-zero operands
# (scripts/micro/mfma_ceiling.hip, profiles/r03_mfma_ceiling.md: 1385-1503 with one wave per SIMD,
# 1615-1626 with two; 2007 on all-zero operands -- the clock follows the power the data draws);
# reported beside the spec fraction, as the copy ceiling is for the HBM kernels
MFMA_SUSTAINED_TFLOPS = 1626.0
# MFMAs a product costs on the split kernels: bf16x3 (three bf16 terms, six of nine cross terms),
# f16x2 (two fp16 terms, three cross terms), f16 (operands rounded to fp16)
SPLIT_MFMAS = (("bf16x3", 6, "bf16 (3-term split of fp32 operands, 6 MFMAs per product, fp32 accumulate)"),
               ("f16x2", 3, "fp16 (2-term split of the power-of-two-scaled fp32 operands, 3 MFMAs per product, fp32 accumulate)"),
               ("_f16_", 1, "fp16 operands (rounded), 1 MFMA per product, fp32 accumulate"))


def synthetic_pair(seed, device):
    """Two independent torch.rand images, ImageNet-normalised (SURVEY.md 8d;
    models/test_models_time.py:17-23, myTransforms/__init__.py:8)."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    left = torch.rand(1, 3, H, W, generator=g)
    right = torch.rand(1, 3, H, W, generator=g)
    return ((left - mean) / std).to(device), ((right - mean) / std).to(device)


def stage_of(kernel):
    """Forward stage a kernel of this library belongs to (per-stage timing)."""
    if kernel.startswith(("conv2d", "spp_", "basicblock2d")):
        return "towers"
    if kernel.startswith("volume"):
        return "volume"
    if kernel.startswith(("soft_argmin", "conv3d_cout1")):
        return "heads"
    return "trunk"


def load_traffic():
    """profiles/traffic.json: HBM-side bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE
    and WRITE_SIZE in separate passes, gfx950 corrections applied), keyed by the plan name of the
    kernel, each entry with the pass and commit it was measured at."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return {}
    with open(tpath) as fh:
        doc = json.load(fh)
    out = {}
    for name, e in doc.get("kernels", {}).items():
        out[name] = e
    return out


def kernel_rooflines(summary, steps):
    """Per kernel: algorithmic work per launch / average launch duration vs the roofline
    that bounds it.  HBM kernels count bytes, MFMA kernels count FLOPs."""
    traffic = load_traffic()
    out = {}
    for name, e in summary.items():
        n, ms, work = e["launches"], e["ms"], e["work"]
        avg_s = ms / n * 1e-3
        per_launch = work / n
        mfma = "mfma" in name
        extra = {}
        split = next((e for e in SPLIT_MFMAS if e[0] in name), None) if mfma else None
        if split is not None:
            # fp32 operands on the 16-bit matrix pipe: `achieved` stays ALGORITHMIC fp32 FLOP/s,
            # `peak` is the dense 16-bit MFMA peak divided by the MFMAs every product costs
            achieved, unit, bound = per_launch / avg_s / 1e12, "TFLOP/s", "mfma"
            peak = round(PEAK_BF16_MFMA_TFLOPS / split[1], 1)
            extra = {"mfma_dtype": split[2], "mfmas_per_product": split[1],
                     "mfma_tflops_executed": round(achieved * split[1], 1),
                     "mfma_peak_16bit": PEAK_BF16_MFMA_TFLOPS,
                     "frac_of_sustained_mfma": round(achieved * split[1] / MFMA_SUSTAINED_TFLOPS, 4),
                     "x_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 3)}
        elif mfma:
            achieved, peak, unit, bound = per_launch / avg_s / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s", "mfma"
        else:
            achieved, peak, unit, bound = per_launch / avg_s / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
        tr = traffic.get(name)
        out[name] = {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                     "frac": round(achieved / peak, 4),
                     "traffic": None if tr is None else tr["bytes_per_launch"],
                     "traffic_source": None if tr is None else tr["source"],
                     "frac_of_copy_ceiling": None if mfma else round(achieved / HBM_COPY_CEILING_GBS, 4),
                     "launches_per_step": n / steps, "avg_launch_us": round(avg_s * 1e6, 2),
                     "ms_per_step": round(ms / steps, 4),
                     "work_per_launch": per_launch, "stage": stage_of(name)}
        out[name].update(extra)
    return out


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            elif int(txt[0]) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    n = min(n, max(1, int(int(txt[0]) / int(fh.read()))))
        except (OSError, ValueError, IndexError):
            pass
    return int(os.environ.get("DSM_BENCH_CPU_THREADS", min(n, 16)))   # 16 = the box's CPU share per GPU


def cpu_baseline(model, left, right, gpu_preds):
    """The oracle restatement of the same forward (same weights, same pair) on the host cores,
    stage by stage (towers / volume / trunk / heads -- models/test_models_time.py:11-32 is the
    reference's own timing script).  Bounded sample: one untimed warm-up pass on a 256x512
    crop, then whole 384x1280 passes until >= 10 s of timed work (at most 3)."""
    from oracle import models as OM          # test infrastructure: baseline + checker only
    from oracle import ops as OO
    torch.set_num_threads(host_cores())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    l, r = left.cpu(), right.cpu()
    net = OM.Net(sd)

    def staged_forward():
        t = [time.perf_counter()]
        fl, fr = OM.psmnet_features(net, l), OM.psmnet_features(net, r)
        t.append(time.perf_counter())
        vol = OO.concat_volume(fl, fr, MAXDISP // 4, mask_left=True)
        t.append(time.perf_counter())
        c1, c2, c3 = OM.psmnet_trunk(net, vol)
        del vol
        t.append(time.perf_counter())
        preds = [OO.soft_argmin(c, (MAXDISP, H, W)) for c in (c3, c2, c1)]
        t.append(time.perf_counter())
        stages = dict(zip(("towers", "volume", "trunk", "heads"),
                          (b - a for a, b in zip(t[:-1], t[1:]))))
        return preds, t[-1] - t[0], stages

    with torch.no_grad():
        OM.forward("psmnet", sd, l[..., :256, :512].contiguous(), r[..., :256, :512].contiguous())
        runs, preds = [], None
        while sum(x[0] for x in runs) < 10.0 and len(runs) < 3:
            preds, total, stages = staged_forward()
            runs.append((total, stages))
    err = max((a.cpu() - b).abs().max().item() for a, b in zip(gpu_preds, preds))
    times = [x[0] for x in runs]
    best, best_stages = min(runs, key=lambda x: x[0])
    return {"value": round(1.0 / best, 5), "unit": "pairs/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d whole forward pass(es) of the same 384x1280 pair, same weights "
                      "(oracle/models.py psmnet stages on torch CPU fp32); value = best of %d; "
                      "%.1f s timed" % (len(times), len(times), sum(times)),
            "seconds_per_pair": {"min": round(best, 3), "median": round(statistics.median(times), 3)},
            "stage_seconds": {k: round(v, 3) for k, v in best_stages.items()}}, err


# ----------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher around it
# ----------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Start n rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set) and
    wait for them.  The parent makes no GPU call: the children are ordinary child processes
    (never an exec of a process that has initialised the GPU).  Rank 0's stdout is the
    parent's, so its JSON line is the line of this command.  Returns the exit code."""
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port()), "DSM_BENCH_SELF_LAUNCHED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL on this host driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code),
                          file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:                      # exactly the processes started above
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the kernel trace of a run (scripts/rocprof_trace.sh) shows the first ~10 replays of the
    # forward speeding up from 5.45 to 5.02 ms while the clocks settle; 30 untimed + 50 timed steps = 0.4 s
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="time eager launches instead of hipGraph replays")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the second, instrumented pass (per-launch HIP events -> rooflines)")
    args = ap.parse_args()

    # DSM_BENCH_REHEARSE=1: rehearsal of the N > 1 control path on a box with fewer GPUs than ranks
    # (ranks share devices, gloo carries the barrier and the max-reduce).  Its value is not a
    # measurement and the JSON line says so.
    rehearse = os.environ.get("DSM_BENCH_REHEARSE") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  The parent makes NO GPU call, not even a device count
        # (torch falls back to hipGetDeviceCount -- hipInit -- when amdsmi discovery fails): a rank
        # without a GPU of its own fails in the child, and that fails the run.  Under a profiler
        # whose preloaded library has already initialised the GPU, starting children is refused.
        if any("rocprofiler" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")):
            raise SystemExit("bench.py --gpus N cannot start its own ranks under a profiler preload: "
                             "profile one rank (--gpus 1) or use torch.distributed.run")
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not rehearse:
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    from dsmnet_amd import calibrate, costvolume
    from dsmnet_amd.models import model_create_by_name
    t_start = time.perf_counter()
    torch.manual_seed(0)
    model = model_create_by_name("psmnet", MAXDISP).to(dev)
    left, right = synthetic_pair(1000 + rank, dev)          # every rank owns its own pair
    calibrate.calibrate_batchnorm(model, left, right)
    calibrate.calibrate_psmnet_heads(model, left, right)
    model.eval()
    torch.cuda.synchronize()
    startup_s = time.perf_counter() - t_start      # model build + BN / head calibration (no MIOpen solver search)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(min(args.warmup, 2)):     # lazily packed weights, allocator: before any capture
            preds = model(left, right)[1]
        # One step = one whole forward.  By default its ~110 launches are replayed from a
        # hipGraph captured once (same kernels, same work; the host launch path leaves the
        # critical path, which keeps 8 ranks on one host from competing for cores).
        step, launch = (lambda: model(left, right)[1]), "eager launches"
        if not args.no_graph:
            try:
                from dsmnet_amd.graphs import GraphedForward
                graphed = GraphedForward(model, left, right)
                step, launch = (lambda: graphed.replay()[1]), "hipGraph replay of the whole forward"
            except Exception as e:                       # capture refused: time eager launches
                print("bench.py: hipGraph capture failed (%s); timing eager launches" % e,
                      file=sys.stderr)
        # the W untimed warm-up steps are steps of the kind that is timed (replays of the captured
        # forward, or eager forwards with --no-graph): r03's kernel trace showed the first ~10 replays
        # after a capture speeding up by 8 % while the clocks settle, which eager warm-up forwards
        # (host-bound, 10 us between launches) do not do for them
        preds = step()
        for _ in range(args.warmup):
            preds = step()
        # the timed region: K steps between the barriers; one HIP event record per step on the
        # launch stream (for the per-step spread) is the only other thing inside
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            preds = step()
            marks[i + 1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        # the same K steps again with two HIP events around every launch of this library (on
        # the launch stream): per-kernel durations for the rooflines.  Kept out of the timed
        # region because the event records themselves cost ~5 % of a step.
        timer = instrumented = None
        volume_b2b_us = volume_seq_us = r01_path_ms = None
        if not args.no_kernel_timing:
            timer = costvolume.LaunchTimer()
            costvolume.set_timer(timer)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model(left, right)
            barrier()
            instrumented = time.perf_counter() - t1
            costvolume.set_timer(None)
            # The cost-volume build (the north-star's >= 60 %-of-HBM kernel).  The default forward
            # never materialises the volume (dres0's first convolution stages it from the split
            # feature maps), so the build is measured (1) inside the forward that DOES materialise it
            # -- the same model with the volume materialised (costvolume option "fuse_volume" off), per-launch
            # HIP events as above -- and (2) launched back to back, where every launch must first
            # drain the previous one's dirty Infinity-Cache lines.
            old_s3 = costvolume.set_option("fuse_volume", False)
            try:
                for _ in range(2):
                    model(left, right)
                vtimer = costvolume.LaunchTimer()
                costvolume.set_timer(vtimer)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(5):
                    model(left, right)
                torch.cuda.synchronize()
                r01_path_ms = (time.perf_counter() - t2) / 5 * 1e3
                costvolume.set_timer(None)
                ve = vtimer.summary().get("volume_ndhwc_fwd_kernel")
                volume_seq_us = ve["ms"] / ve["launches"] * 1e3 if ve else None
            finally:
                costvolume.set_timer(None)
                costvolume.set_option("fuse_volume", old_s3)
            # back to back: 20 launches replayed from one hipGraph (no host launch path between them)
            fl, fr = model.features(left, right)
            fl, fr = fl.contiguous(), fr.contiguous()
            nrep = 20
            costvolume.concat_volume(fl, fr, MAXDISP // 4, True)
            torch.cuda.synchronize()
            vgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(vgraph):
                vols = [costvolume.concat_volume(fl, fr, MAXDISP // 4, True) for _ in range(2)]
                for _ in range(nrep - 2):
                    costvolume.concat_volume(fl, fr, MAXDISP // 4, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            vgraph.replay()
            torch.cuda.synchronize()
            e0.record()
            vgraph.replay()
            e1.record()
            torch.cuda.synchronize()
            volume_b2b_us = e0.elapsed_time(e1) / nrep * 1e3
            del vols, vgraph
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": METRIC, "value": round(world * args.steps / elapsed, 3), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, not a measurement)" if rehearse else ""),
            "config": {"workload": "PSMNet stacked-hourglass forward, D=192, 384x1280, one pair "
                                   "per GPU per step (BASELINE configs[3]: batch 8 over 8 GPUs)",
                       "height": H, "width": W, "maxdisp": MAXDISP, "pairs_per_gpu_per_step": 1,
                       "heads": 3, "parallelism": "pairs sharded over ranks, no collective",
                       "inputs": "left and right: independent torch.rand images, ImageNet-normalised",
                       "weights": "reference init (seed 0), BN + heads calibrated",
                       "launch": launch,
                       "launcher": ("self-launched child ranks" if os.environ.get("DSM_BENCH_SELF_LAUNCHED")
                                    else ("external launcher" if world > 1 else "single process")),
                       "conv_precision": costvolume.get_option("conv_precision") +
                                         " (costvolume option conv_precision / DSM_CONV_PRECISION: f16x2 | bf16x3 | fp32 are "
                                         "fp32-accurate, f16 is the reduced-precision mode)",
                       "trunk_path": "z-sliding kernel for the 32-channel stride-1 layers, cost volume "
                                     "never materialised"},
            # rank 0's per-step GPU time from one HIP event per step inside the timed region
            "startup_s": round(startup_s, 2),
            "step_ms": {"median": round(statistics.median(step_ms), 3), "min": round(min(step_ms), 3),
                        "max": round(max(step_ms), 3), "mean_wall": round(ms_per_step, 3)},
        }
        if timer is not None:
            roofs = kernel_rooflines(timer.summary(), args.steps)
            dominant = max(roofs, key=lambda k: roofs[k]["ms_per_step"])
            result["roofline"] = dict(roofs[dominant], kernel=dominant)
            if volume_b2b_us and volume_seq_us:
                vbytes = 4.0 * (2 * 32 * (H // 4) * (W // 4) + 64 * (MAXDISP // 4) * (H // 4) * (W // 4))
                result["cost_volume_build"] = {
                    "kernel": "volume_ndhwc_fwd_kernel", "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                    "algorithmic_bytes": vbytes,
                    "in_forward_us": round(volume_seq_us, 2),
                    "achieved": round(vbytes / (volume_seq_us * 1e-6) / 1e9, 1),
                    "frac": round(vbytes / (volume_seq_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                    "frac_of_copy_ceiling": round(vbytes / (volume_seq_us * 1e-6) / 1e9 / HBM_COPY_CEILING_GBS, 4),
                    "back_to_back_us": round(volume_b2b_us, 2),
                    "back_to_back_frac": round(vbytes / (volume_b2b_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                    "how": "the default forward does not launch this kernel (the volume is staged from "
                           "the towers' output inside dres0's first convolution); in_forward_us = its "
                           "average over 5 eager forwards of the SAME model with the volume materialised "
                           "(costvolume option fuse_volume off), HIP events on the launch stream; "
                           "back_to_back_us = 20 launches replayed from one hipGraph (every launch must first "
                           "drain the previous one's dirty Infinity-Cache lines)",
                    "materialising_path_ms_per_step_eager": round(r01_path_ms, 3)}
            result["rooflines"] = roofs
            hip_ms = sum(v["ms_per_step"] for v in roofs.values())
            result["hip_path_ms_per_step"] = round(hip_ms, 3)
            stages = {}
            for v in roofs.values():
                stages[v["stage"]] = round(stages.get(v["stage"], 0.0) + v["ms_per_step"], 4)
            result["stage_ms_per_step"] = stages
            # stock torch ops, launch gaps and the event records themselves, in the instrumented pass
            result["other_ms_per_step"] = round(instrumented / args.steps * 1e3 - hip_ms, 3)
            result["kernel_timing"] = {
                "how": "%d eager steps right after the timed region with two HIP events per "
                       "launch on the launch stream" % args.steps,
                "instrumented_ms_per_step": round(instrumented / args.steps * 1e3, 3)}
        if world == 1 and not args.no_cpu_baseline:
            base, err = cpu_baseline(model, left, right, preds)
            result["cpu_baseline"] = base
            result["parity_max_abs_px_vs_cpu"] = err
        result["precision"] = {
            "storage_and_accumulate": "f32",
            "conv_products": {
                "fp32": "fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products and sums",
                "bf16x3": "each fp32 operand split exactly into 3 bf16 terms; 6 bf16 MFMAs per product, "
                          "dropped terms <= 3*2^-25 relative; fp32 accumulate",
                "f16x2": "each fp32 operand scaled by a power of two (from the tensor's device-side absolute "
                         "maximum) and split into 2 fp16 terms (22 significand bits); 3 fp16 MFMAs per product; "
                         "fp32 accumulate",
                "f16": "operands rounded to fp16 after the power-of-two scaling; 1 MFMA per product; fp32 "
                       "accumulate (reduced precision: BASELINE config #5)"}[costvolume.get_option("conv_precision")],
            "conv_error_vs_float64": "max rel / rms on 8 layer shapes, inputs 1e-9 .. 1e7 (scripts/precision_check.py, "
                                     "profiles/r03_precision.md): fp32-input MFMA 0.8-1.5e-6 / 4.2-5.9e-7, bf16x3 "
                                     "0.8-1.5e-6 / 4.1-5.9e-7, f16x2 0.6-1.1e-6 / 3.2-4.4e-7, f16 3e-4 / 2.9e-4",
            "gate": "forward disparity vs the fp32 CPU reference path <= 1e-3 px (parity_max_abs_px_vs_cpu)"}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
