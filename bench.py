#!/usr/bin/env python3
"""Throughput benchmark of the ORB extraction hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one full extraction (pyramid, FAST + quadtree, orientation, blur + rBRIEF) of a batch
of synthetic 1920x1080 frames that already sit in HBM (workload C4: 2000 features, 8 levels, scale
1.2, FAST 20/7; 1024 frames per GPU and step, which the library runs as two passes of 512). Frames
are independent: every rank owns a contiguous range of them and no collective runs on the data path;
the job value is total frames / max rank time. Rank 0 prints one JSON line.
  --scaling weak   (default) every rank extracts --batch frames per step
  --scaling strong the job's --total-batch frames (or pairs) are cut into one contiguous range per rank
  --preset c4_strong_1024 / c3_stereo_512 : the two sharded configurations of BASELINE.json by name

The line also carries
  roofline     : the kernel class with the largest device time, its algorithmic bytes per launch
                 divided by its average launch duration over the timed steps (HIP events recorded
                 inside libORBhip on the streams the kernels run on), against the 8 TB/s HBM3E peak;
  cpu_baseline : the CPU oracle (a restatement of the reference's CPU extractor) timed on a
                 bounded sample of the same workload on the host cores of this box.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak
VALU_CYCLES_PER_INST = 4.25   # measured issue cost of the instruction kinds the kernels use (profiles/r02_valu_calib.txt)

WORKLOADS = {
    # name: (w, h, nfeatures, scale, levels, iniTh, minTh)
    "C4_1080p_2000f": (1920, 1080, 2000, 1.2, 8, 20, 7),
    "C5_4k_8000f": (3840, 2160, 8000, 1.1, 12, 20, 7),
    "C1_tum": (640, 480, 1000, 1.2, 8, 20, 7),
    "C2_kitti": (1241, 376, 2000, 1.2, 8, 20, 7),
    "C3_euroc": (752, 480, 1200, 1.2, 8, 20, 7),
}


def level_sizes(w, h, scale, levels):
    """Level sizes exactly as ComputePyramid derives them (float32 scale chain, round half even)."""
    sf = np.float32(1.0)
    out = []
    for l in range(levels):
        if l:
            sf = np.float32(np.float64(sf) * np.float64(np.float32(scale)))
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algorithmic_bytes(w, h, scale, levels, n_out):
    """Bytes per frame a perfect implementation must move, split by kernel class.

    total = W*H + sum (w_l+38)(h_l+38) + sum_{l<n-1} w_l*h_l + 2*sum w_l*h_l + 60*N_out
    """
    lv = level_sizes(w, h, scale, levels)
    area = [a * b for a, b in lv]
    bordered = [(a + 38) * (b + 38) for a, b in lv]
    parts = {
        "k_level0": w * h + bordered[0],
        "k_resize": sum(area[:-1]) + sum(bordered[1:]),
        "k_fast": sum(area),
        "k_quadtree": 0,
        "k_describe": sum(area) + 60 * n_out,
    }
    return parts, sum(parts.values())


def compulsory_bytes(w, h, scale, levels, n_out):
    """Stricter lower bound: input read + bordered pyramid write + outputs (no intermediate re-reads)."""
    lv = level_sizes(w, h, scale, levels)
    return w * h + sum((a + 38) * (b + 38) for a, b in lv) + 60 * n_out


def device_copy_gbs(torch, device, nbytes=1 << 30, reps=5):
    """Measured device-to-device copy bandwidth (read + write bytes per second) of this GPU."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (time.perf_counter() - t0) / 1e9


def profiled_traffic(kernel, frames):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary
    (profiles/*_traffic.json, FETCH_SIZE/WRITE_SIZE passes), scaled to `frames` per launch."""
    files = sorted((ROOT / "profiles").glob("*_traffic.json"))
    if not files:
        return None, None
    try:
        t = json.loads(files[-1].read_text())
        k = t["kernels"][kernel]
        return int(k["hbm_bytes_per_launch"] * frames / t["frames_per_launch"]), files[-1].name
    except (KeyError, ValueError):
        return None, None


def profiled_valu(kernel, frames):
    """Vector-ALU wave instructions per launch of `kernel` from the newest committed SQ counter
    summary (profiles/*_valu.json, tools/pmc_valu.py), scaled to `frames` per launch."""
    files = sorted((ROOT / "profiles").glob("*_valu.json"))
    if not files:
        return None, None
    try:
        t = json.loads(files[-1].read_text())
        return int(t["kernels"][kernel]["valu_insts_per_launch"] * frames / t["frames_per_launch"]), files[-1].name
    except (KeyError, ValueError):
        return None, None


def make_batch_on_device(torch, kind, w, h, nframes, device, nbase=16, start=0):
    """nbase host-generated frames (synthetic frame indices from `start` on), expanded on the device by
    distinct circular shifts."""
    from visual_slam_amd import synth
    nbase = min(nbase, nframes)
    base = torch.from_numpy(synth.batch(kind, w, h, nbase, start)).to(device)
    out = torch.empty((nframes, h, w), dtype=torch.uint8, device=device)
    for i in range(nframes):
        b, k = i % nbase, i // nbase
        out[i] = base[b] if k == 0 else torch.roll(base[b], shifts=(37 * k % h, 101 * k % w), dims=(0, 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="frames (pairs with --stereo) per GPU per step in weak scaling")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch frames per GPU; strong: --total-batch frames over all GPUs (contiguous range per rank)")
    ap.add_argument("--total-batch", type=int, default=None, help="frames (pairs with --stereo) per step of the whole job in strong scaling")
    ap.add_argument("--preset", default=None, choices=["c4_strong_1024", "c3_stereo_512"],
                    help="c4_strong_1024: 1024 1080p frames per step over all GPUs; c3_stereo_512: 512 EuRoC-shaped stereo pairs per step over all GPUs")
    ap.add_argument("--max-batch", type=int, default=0, help="frames per internal pass (0 = library default)")
    ap.add_argument("--workload", default="C4_1080p_2000f", choices=sorted(WORKLOADS))
    ap.add_argument("--kind", default="blocks", choices=["blocks", "noise", "lowc"])
    ap.add_argument("--cpu-frames", type=int, default=-1,
                    help="frames of the CPU baseline sample (0 = skip; default: the pixels of 256 1080p frames, at most 256 frames: 20-30 s of CPU work)")
    ap.add_argument("--stereo", action="store_true", help="step = left + right extraction + stereo matching of every pair; value = pairs/s")
    ap.add_argument("--rectify", action="store_true", help="rectify every frame through cv::remap-style maps on the way into level 0")
    ap.add_argument("--color", type=int, default=0, choices=[0, 3, 4], help="feed interleaved BGR(A) frames: gray conversion fused into level 0")
    ap.add_argument("--host-fed", action="store_true", help="also time the PCIe-inclusive host-buffer entry point")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for the barrier / max-time reduction")
    ap.add_argument("--fast-th", default=None, help="INI,MIN: other FAST thresholds than the workload's (an experiment, named in config.fast_th)")
    ap.add_argument("--skip-alone", action="store_true", help="do not time the kernels one at a time after the timed steps (profile runs)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a one-GPU box: every rank uses GPU 0 (use with --backend gloo)")
    args = ap.parse_args()
    if args.preset == "c4_strong_1024":
        args.workload, args.scaling, args.stereo = "C4_1080p_2000f", "strong", False
        args.total_batch = args.total_batch or 1024     # (--total-batch n: what one of 1024 / n ranks sees, on one GPU)
    elif args.preset == "c3_stereo_512":
        args.workload, args.scaling, args.stereo = "C3_euroc", "strong", True
        args.total_batch = args.total_batch or 512
    args.total_batch = args.total_batch or 1024

    import torch
    from visual_slam_amd import orbhip, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libORBhip has no CPU path)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    w, h, nfeat, scale, levels, ini_th, min_th = WORKLOADS[args.workload]
    if args.fast_th:
        ini_th, min_th = (int(v) for v in args.fast_th.split(","))
    ex = orbhip.ORBextractor(nfeat, scale, levels, ini_th, min_th, device=local_rank, max_batch=args.max_batch)
    cap = ex.max_keypoints()
    if args.rectify:
        ex.set_rectification(*synth.rectify_maps(w, h))
    if args.scaling == "strong":
        from visual_slam_amd import sharding
        lo, hi = sharding.shard_range(args.total_batch, world, rank)     # a unit is a frame, or a pair with --stereo
        B = hi - lo
        if B < 1:
            sys.exit("--total-batch %d leaves rank %d without work" % (args.total_batch, rank))
        job_units = args.total_batch
    else:
        B = args.batch
        job_units = world * B
    frames = make_batch_on_device(torch, args.kind, w, h, B, device, start=16 * rank)   # every rank its own frames
    d_kp = torch.empty((B, cap, 7), dtype=torch.float32, device=device)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=device)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=device)
    stream = torch.cuda.Stream(device)   # one explicit stream for every library call of this rank

    if args.stereo:
        # right image of every pair: the left image moved 12 px to the left (KITTI 00-02 calibration)
        exr = orbhip.ORBextractor(nfeat, scale, levels, ini_th, min_th, device=local_rank, max_batch=args.max_batch)
        frames_r = torch.roll(frames, shifts=-12, dims=2).contiguous()
        r_kp, r_desc, r_cnt = torch.empty_like(d_kp), torch.empty_like(d_desc), torch.zeros_like(d_cnt)
        d_ur = torch.empty((B, cap), dtype=torch.float32, device=device)
        d_dep = torch.empty((B, cap), dtype=torch.float32, device=device)
        d_nm = torch.zeros((B,), dtype=torch.int32, device=device)
        fx, bf = 718.856, 386.1448

    if args.color:
        # gray replicated into every channel: the conversion returns the same gray image
        # (the weights sum to 2^14), so keypoint counts and later kernels stay comparable
        frames_c = frames.unsqueeze(-1).expand(B, h, w, args.color).contiguous()

    def step():
        if args.stereo:
            # both extractions and the matching, pass by pass inside the library (any batch size)
            orbhip.extract_stereo_batch_device(ex, exr, frames.data_ptr(), frames_r.data_ptr(), B, w, h, w, w * h,
                                               d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(),
                                               r_kp.data_ptr(), r_desc.data_ptr(), r_cnt.data_ptr(), cap, bf / fx, bf,
                                               d_ur.data_ptr(), d_dep.data_ptr(), d_nm.data_ptr(), stream.cuda_stream)
            return
        if args.color:
            ex.extract_batch_device_color(frames_c.data_ptr(), B, w, h, w * args.color, w * h * args.color, args.color, 0,
                                          d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, stream.cuda_stream)
        else:
            ex.extract_batch_device(frames.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(),
                                    d_cnt.data_ptr(), cap, stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()             # inputs were produced on torch's default stream
    for _ in range(args.warmup):
        step()
    barrier()
    # per-kernel device time of the timed steps themselves: HIP events recorded inside libORBhip
    # on the streams the kernels run on (two event records per launch, no host synchronisation)
    ex.enable_timing(True)
    if args.stereo:
        exr.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ex.synchronize()   # raises on a deferred capacity error
    timing = ex.timing()
    busy = ex.timing_busy()
    ex.enable_timing(False)
    if args.stereo:
        exr.synchronize()
        tr = exr.timing()
        br = exr.timing_busy()
        exr.enable_timing(False)
        timing = {k: (timing[k][0] + tr[k][0], timing[k][1] + tr[k][1]) for k in timing}
        busy = {k: busy[k] + br[k] for k in busy}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fps = job_units * args.steps / dt

    if rank == 0:
        counts = d_cnt.cpu().numpy()
        n_out = float(counts.mean())
        parts, total_bytes = algorithmic_bytes(w, h, scale, levels, nfeat)
        if args.stereo:
            # per pair: both extractions + keypoint/descriptor records of both sides read, two floats per left keypoint written
            parts = {k: 2 * v for k, v in parts.items()}
            parts["k_stereo"] = 2 * 60 * nfeat + 8 * nfeat
            total_bytes = sum(parts.values())
        else:
            parts["k_stereo"] = 0
        # (busy: the time during which at least one launch of the class ran; launches of a class that run side by side
        # on two streams count once)
        kname = max(busy, key=lambda k: busy[k])
        k_ms, k_launches = busy[kname], timing[kname][1]
        lanes = max(1, min(int(os.environ.get("ORBHIP_LANES", "1")), 4))
        passes = max(timing["k_quadtree"][1] // (2 if args.stereo else 1), 1)   # one quadtree launch per pass and side
        frames_per_launch = B * args.steps / passes
        # A "launch" below is what a kernel class launches for one pass (k_fast: the plain tiles level by level on the
        # second stream and the packed tiles; k_resize: the seven levels of the chain): bytes and time of a pass.
        k_bytes_per_launch = parts[kname] * frames_per_launch
        k_avg_s = k_ms * 1e-3 / passes
        achieved = k_bytes_per_launch / k_avg_s / 1e9
        # the committed counter summaries were collected on the default workload: they say nothing about another one
        profiled = (args.workload == "C4_1080p_2000f" and args.kind == "blocks" and not args.stereo and not args.rectify
                    and not args.color)
        traffic, traffic_src = profiled_traffic(kname, frames_per_launch) if profiled else (None, None)
        copy_gbs = device_copy_gbs(torch, device)
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": int(k_bytes_per_launch), "avg_launch_ms": round(k_avg_s * 1e3, 4),
                    "launch_is": "the launches of the kernel class for one pass (%.1f timed spans per pass; time = that during which at least one of them ran)" % (k_launches / passes),
                    "timed": "as run",
                    "whole_path_bytes_per_frame": int(total_bytes),
                    "whole_path_frac": round(B * args.steps / dt * total_bytes / 1e9 / HBM_PEAK_GBS, 4),
                    "compulsory_bytes_per_frame": int(compulsory_bytes(w, h, scale, levels, nfeat)),
                    "device_copy_GBs": round(copy_gbs, 1),
                    "whole_path_frac_of_copy": round(B * args.steps / dt * total_bytes / 1e9 / copy_gbs, 4),
                    "frames_per_pass": int(frames_per_launch), "passes_in_flight": lanes,
                    "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in busy.items()}}
        # the limit that actually binds the dominant kernel: vector-ALU issue. Measured on this chip
        # (tools/calib/valu_calib.hip, profiles/r02_valu_calib.txt): with 2..8 waves per SIMD a SIMD needs
        # 4.25 cycles per wave64 instruction for packed 16-bit, 3-operand, DPP, SDWA, multiply and
        # compare instructions (what these kernels are made of) and 2.2 for plain 32-bit add / sub /
        # and / or / shift-right / move streams. The fraction below prices every instruction at 4.25 cycles
        # at the 2.4 GHz peak clock.
        valu, valu_src = profiled_valu(kname, frames_per_launch) if profiled else (None, None)
        if valu:
            # Priced per instruction class it cannot pass 1: every wave64 instruction needs at least the 4 cycles of a
            # 16-lane SIMD except the plain two-operand 32-bit / 16-bit class, which the calibration measured at 2.2.
            # The counters give no split by class, so the ceiling used is the architectural one, 4 cycles for every
            # instruction ("frac": an upper estimate of the busy share when part of the stream is in the fast class);
            # "frac_at_calibrated_4.25" is the same count against the measured rate of the slow class, and can pass 1
            # by exactly the share of fast-class instructions.
            peak_arch = 256 * 4 / 4.0 * 2.4e9
            peak_cal = 256 * 4 / VALU_CYCLES_PER_INST * 2.4e9
            roofline["valu"] = {"wave_insts_per_launch": valu, "source": valu_src, "achieved_insts_per_s": round(valu / k_avg_s, 1),
                                "peak_insts_per_s": peak_arch, "cycles_per_inst_peak": 4.0, "frac": round(valu / k_avg_s / peak_arch, 4),
                                "cycles_per_inst_calibrated": VALU_CYCLES_PER_INST, "calibration": "profiles/r02_valu_calib.txt",
                                "frac_at_calibrated_4.25": round(valu / k_avg_s / peak_cal, 4)}
        # By default FAST runs on a second stream beside the resize chain, level by level (ORBHIP_OVERLAP, large passes),
        # and ORBHIP_LANES > 1 keeps several passes in flight: the launch durations above are then those of kernels that
        # share the chip. The same kernels are then also timed one at a time (a second handle with both switched off,
        # one untimed and two timed steps): a kernel with the GPU to itself. That is the figure the roofline object
        # leads with (it says what the kernel achieves, and it is what a kernel trace of ORBHIP_OVERLAP=0 shows); the
        # figures as run move to "as_run".
        ov = os.environ.get("ORBHIP_OVERLAP", "1")
        overlap_on = ov == "2" or (ov != "0" and frames_per_launch * w * h >= (1.0e8 if args.stereo else 5.0e7))   # the library's own rule
        roofline["fast_beside_resize"] = bool(overlap_on)
        if (lanes > 1 or overlap_on) and not args.skip_alone and not args.stereo:   # (the stereo step has no one-at-a-time twin here)
            saved = {k: os.environ.get(k) for k in ("ORBHIP_LANES", "ORBHIP_OVERLAP")}
            os.environ["ORBHIP_LANES"] = "1"
            os.environ["ORBHIP_OVERLAP"] = "0"
            ex1 = orbhip.ORBextractor(nfeat, scale, levels, ini_th, min_th, device=local_rank, max_batch=args.max_batch)
            if args.rectify:
                ex1.set_rectification(*synth.rectify_maps(w, h))
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k)
                else:
                    os.environ[k] = v
            def step1():
                if args.color:
                    ex1.extract_batch_device_color(frames_c.data_ptr(), B, w, h, w * args.color, w * h * args.color, args.color, 0,
                                                   d_kp.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, stream.cuda_stream)
                else:
                    ex1.extract_batch_device(frames.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(),
                                             d_cnt.data_ptr(), cap, stream.cuda_stream)
            step1()
            torch.cuda.synchronize()
            ex1.enable_timing(True)
            step1(); step1()
            torch.cuda.synchronize()
            t1 = ex1.timing()
            ex1.enable_timing(False)
            s_ms, s_launches = t1[kname]
            s_passes = max(t1["k_quadtree"][1], 1)
            s_bytes = parts[kname] * B * 2 / s_passes
            s_avg = s_ms * 1e-3 / s_passes
            alone = {"achieved": round(s_bytes / s_avg / 1e9, 1), "frac": round(s_bytes / s_avg / 1e9 / HBM_PEAK_GBS, 4),
                     "bytes_per_launch": int(s_bytes), "avg_launch_ms": round(s_avg * 1e3, 4),
                     "kernel_ms_per_step": {k: round(v[0] / 2, 3) for k, v in t1.items()},
                     "ms_per_step": round(sum(v[0] for v in t1.values()) / 2, 3)}
            roofline["as_run"] = {k: roofline[k] for k in ("achieved", "frac", "avg_launch_ms", "launch_is", "kernel_ms_per_step")}
            roofline["as_run"]["note"] = ("kernels share the chip: FAST beside the resize chain" +
                                          (", %d passes in flight" % lanes if lanes > 1 else ""))
            roofline.update({"achieved": alone["achieved"], "frac": alone["frac"], "avg_launch_ms": alone["avg_launch_ms"],
                             "kernel_ms_per_step": alone["kernel_ms_per_step"], "kernel_ms_sum_per_step": alone["ms_per_step"],
                             "timed": "one kernel at a time (a second handle with ORBHIP_OVERLAP=0 and ORBHIP_LANES=1 in this process, "
                                      "HIP events on the kernels' stream, two steps)",
                             "launch_is": "the launches of the kernel class for one pass (%.1f per pass), one after the other" % (s_launches / s_passes)})
            if "valu" in roofline:
                v = roofline["valu"]
                v["as_run_frac"] = v["frac"]
                v["achieved_insts_per_s"] = round(v["wave_insts_per_launch"] / s_avg, 1)
                v["frac"] = round(v["wave_insts_per_launch"] / s_avg / v["peak_insts_per_s"], 4)
                v["frac_at_calibrated_4.25"] = round(v["wave_insts_per_launch"] / s_avg / (256 * 4 / VALU_CYCLES_PER_INST * 2.4e9), 4)
            del ex1
        cpu = None
        if args.cpu_frames < 0:
            args.cpu_frames = max(16, min(256, int(256 * (1920 * 1080) / (w * h))))
        if args.cpu_frames > 0 and world == 1:           # the CPU baseline is timed at N = 1 only
            from oracle import orb_oracle   # checker / baseline only, never on the measured GPU path
            from visual_slam_amd import synth
            # one oracle instance per host core this process may run on (the affinity mask; os.cpu_count() would
            # name every CPU of the machine, most of which a one-GPU share of the box does not own)
            host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            nthreads = max(1, host_cores)
            if args.stereo:
                # pairs dealt to worker threads (ctypes releases the GIL), each with its own two oracle instances
                from concurrent.futures import ThreadPoolExecutor
                npairs = max(min(args.cpu_frames // 2, B), 1)
                hl, hr = frames[:npairs].cpu().numpy(), frames_r[:npairs].cpu().numpy()   # the first pairs of the GPU batch
                pairs = list(zip(hl, hr))
                prm = (nfeat, scale, levels, ini_th, min_th)

                def work(t):
                    eL, eR = orb_oracle.OracleExtractor(*prm), orb_oracle.OracleExtractor(*prm)
                    for left, right in pairs[t::nthreads]:
                        kl, dl = eL.extract(left)
                        kr, dr = eR.extract(right)
                        orb_oracle.stereo_match(eL, eR, kl, dl, kr, dr, bf / fx, bf)
                t1 = time.perf_counter()
                with ThreadPoolExecutor(nthreads) as pool:
                    list(pool.map(work, range(nthreads)))
                secs = time.perf_counter() - t1
                cpu = {"value": round(npairs / secs, 2), "unit": "pairs/s", "cores": nthreads, "threads": nthreads, "host_cores": host_cores,
                       "machine_cpus": os.cpu_count(), "kind": "port",
                       "sample": "the first %d of the %d %s pairs of %s the GPU step extracts, two oracle instances per thread" % (npairs, B, args.kind, args.workload)}
            else:
                ncpu = min(args.cpu_frames, B)
                sample = frames[:ncpu].cpu().numpy()     # the first frames of the very batch the GPU steps extract
                secs, ccounts = orb_oracle.extract_batch_mt((nfeat, scale, levels, ini_th, min_th), sample, nthreads)
                same = bool(np.array_equal(ccounts, counts[:ncpu]))
                cpu = {"value": round(ncpu / secs, 2), "unit": "frames/s", "cores": nthreads, "threads": nthreads, "host_cores": host_cores,
                       "machine_cpus": os.cpu_count(), "kind": "port",
                       "sample": "the first %d of the %d %s frames of %s the GPU step extracts, one oracle instance per thread" % (ncpu, B, args.kind, args.workload),
                       "keypoint_counts_equal_gpu": same}
        extra = {}
        if args.host_fed:
            # PCIe-inclusive rate of the host entry point (images in host memory in, results in host memory out): never
            # `value`. Pageable images are packed into page-locked staging by the library's host threads, page-locked
            # ones are copied from where they lie. Beside it the host-to-device copy bandwidth measured here.
            nh = min(B, 512)
            pageable = frames[:nh].cpu()
            pinned = pageable.pin_memory()
            dst = torch.empty_like(frames[:nh])
            dst.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                dst.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
            h2d_gbs = 3 * pinned.numel() / (time.perf_counter() - t1) / 1e9
            h_kp = np.zeros((nh, cap), orbhip.KP_DTYPE)
            h_desc = np.zeros((nh, cap, 32), np.uint8)
            h_cnt = np.zeros(nh, np.int32)
            hf = {"frames": nh, "h2d_copy_GBs": round(h2d_gbs, 1), "h2d_ceiling_frames_per_s": round(h2d_gbs * 1e9 / (w * h), 1)}
            for name, t in (("pageable", pageable), ("pinned", pinned)):
                arr = t.numpy()
                best = 0.0
                for rep in range(3):                       # the first call builds the plan and the staging buffers
                    t1 = time.perf_counter()
                    rc = orbhip.lib().orbhip_extract_batch(ex._h, arr.ctypes.data, nh, w, h, w, w * h, h_kp.ctypes.data,
                                                           h_desc.ctypes.data, h_cnt.ctypes.data, cap)
                    secs = time.perf_counter() - t1
                    if rc:
                        raise orbhip.OrbHipError(rc, orbhip.lib().orbhip_last_error(ex._h).decode())
                    if rep:
                        best = max(best, nh / secs)
                hf[name] = {"frames_per_s": round(best, 1), "image_GBs": round(best * w * h / 1e9, 2),
                            "frac_of_h2d_copy": round(best * w * h / 1e9 / h2d_gbs, 3),
                            "counts_equal_device_resident": bool(np.array_equal(h_cnt, counts[:nh]))}
            extra["host_fed"] = hf
        line = {
            "metric": ("ORB extract + stereo match pairs/s" if args.stereo else
                       "ORB extract frames/s (1080p, 2k feats)" if args.workload == "C4_1080p_2000f" else "ORB extract frames/s"),
            "value": round(fps, 1), "unit": "pairs/s" if args.stereo else "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload, "image": [w, h], "nfeatures": nfeat, "levels": levels, "scale": scale,
                       "fast_th": [ini_th, min_th], "frames_per_gpu_per_step": B, "frames_per_step": job_units, "distribution": args.kind,
                       "base_frames": min(16, B), "expanded_by": "circular shifts of the base frames (37 rows, 101 columns per repeat)",
                       "inputs": "device-resident" + (", %d-channel BGR" % args.color if args.color else "") + (", rectified by remap" if args.rectify else ""), "pyramid_copied_back": False, "blur_variant": "legacy",
                       "mean_keypoints_per_frame": round(n_out, 1), "sharding": "frames, no collective",
                       **({"stereo": True, "mean_matches_per_pair": round(float(d_nm.float().mean().item()), 1)} if args.stereo else {})},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
