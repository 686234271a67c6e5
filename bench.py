"""Headline benchmark: frames/s of the per-frame hot path (224x224 crops -> SPIN ResNet-50 encoder +
regressor -> rotation conversions -> SMPL LBS (mesh + joints) -> REBA/RULA) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 64 synthetic crops per GPU (BASELINE config 2:
"Batch=64 random 224x224 crops, ResNet-50+SMPL fp32").  Frames shard across GPUs (weak scaling);
with N > 1 every step's per-frame SMPL-parameter record is all-gathered over RCCL on a side stream.
Prints ONE JSON line on rank 0.

    configs[3]'s per-GPU workload (2048 frames on 8 GPUs = 256 per GPU):  ... bench.py --gpus N --batch 256 --lanes 2
    configs[2] (bf16 encoder):                                            python bench.py --precision bf16 --batch 256 --lanes 2
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense
CONV_GFLOP_PER_FRAME = 8.174272512  # 4 087 136 256 MAC
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E
SMPL_CONST_BYTES = 19_350_000      # model constants read once per launch
SMPL_BYTES_PER_FRAME = 83_296      # pose+betas in, verts+joints out
SMPL_FLOP_PER_FRAME = 2 * 7_737_470  # SMPL MACs per frame (packed-FMA VALU peak = 157.3 TFLOP/s too)


def host_cores():
    """CPU threads this process may really use: scheduler affinity, capped by the cgroup quota and by
    the GPU box's per-GPU CPU share (16) -- os.cpu_count() reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(sd, sm, info, frames):
    """The oracle ("port") arranged as the reference runs it (base.py:211-240), timed on host cores."""
    from oracle import hmr_ref, pipeline_ref, smpl_ref
    from poserisk_release_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    ref = hmr_ref.build(sd)
    om = smpl_ref.SMPLModel(sm["v_template"], sm["shapedirs"], sm["posedirs"], sm["J_regressor"], sm["weights"])
    x = synth.crops(frames + 8, seed=123)
    pipeline_ref.run(ref, om, x[:8], info, per_frame_scorers=True)            # warm-up
    t = {}
    t0 = time.perf_counter()
    pipeline_ref.run(ref, om, x[8:], info, timings=t, per_frame_scorers=True)
    dt = time.perf_counter() - t0
    return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frames, encoder batch 8 on torch-CPU fp32, per-frame Rodrigues/Euler loops, "
                      f"batch-1 SMPL per frame, REBA+RULA frame by frame (the oracle's scorers called per frame, as the "
                      f"reference's classes loop over poses); stage seconds "
                      + ", ".join(f"{k}={v:.2f}" for k, v in t.items())}


def measure_other_config(precision, B, lanes, steps, warmup, dev, sd, sm, info, workload):
    """One more configuration in the same process (fresh handles): `value` over EXACTLY `steps` steps between
    synchronisations with `lanes` batches in flight, then the same steps with one batch in flight and every conv launch
    bracketed by hipEvents -> roofline figures (as the headline's, compact)."""
    from poserisk_release_amd import pipeline as pl
    from poserisk_release_amd.hmr import HMR
    from poserisk_release_amd.smpl_layer import SMPLLayer
    model = HMR(max_batch=B, precision=precision).to(dev)
    model.load_state_dict(sd)
    layer = SMPLLayer(sm, device=dev, max_batch=max(B, 16))
    pipe = pl.FramePipeline(model, layer, info, with_verts=True, lanes=lanes)
    pipe.prepare(B, dev)
    gen = torch.Generator(device=dev).manual_seed(2000)
    crops = torch.rand((B, 3, 224, 224), generator=gen, device=dev, dtype=torch.float32)

    def fence():
        pipe.synchronize()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        pipe(crops)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe(crops)
    fence()
    elapsed = time.perf_counter() - t0
    prof = pl.FramePipeline(model, layer, info, with_verts=True, lanes=1)
    for _ in range(2):
        prof(crops)
    torch.cuda.synchronize(dev)
    model.profile_enable(True)
    for _ in range(steps):
        prof(crops)
    torch.cuda.synchronize(dev)
    ms, cnt, flops_per_frame, mfma_flops_per_frame = model.profile_read(with_mfma_flops=True)
    model.profile_enable(False)
    peak = PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    conv_s = float(ms.sum()) * 1e-3
    achieved = float(flops_per_frame.sum()) * B * steps / conv_s / 1e12
    executed = float(mfma_flops_per_frame.sum()) * B * steps / conv_s / 1e12
    out = {"workload": workload, "dtype": "f32" if precision == "fp32" else "bf16 (encoder; f32 accumulate, regressor/SMPL f32)",
           "frames_per_step": B, "batches_in_flight": lanes, "value": round(steps * B / elapsed, 1), "unit": "frames/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
           "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "mfma_executed_frac": round(executed / peak, 4),
                        "conv_ms_per_step": round(conv_s / steps * 1e3, 4), "conv_launches_per_step": int(cnt.sum()) // steps,
                        "conv_kernels_per_step": sum(model.plan_counts(B)[i] * m for i, m in ((0, 1), (1, 2))),
                        "batches_in_flight": 1}}
    del prof, pipe, model, layer, crops
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a K-step region starts from an idle GPU and ends with a drain of the batches in flight, which costs a short
    # region more than a long one (same box: 30 steps read 16.9 - 17.0 k frames/s, 100 read 17.15 - 17.2 k, 300 read 17.29 k)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=1024, help="frames in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--streams", type=int, default=0, help="encoder sub-batch streams inside one batch (0 = library default 1)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="encoder precision: fp32 = configs[1] (headline), bf16 = configs[2] (use --batch 256)")
    ap.add_argument("--lanes", type=int, default=3, help="whole batches in flight on separate HIP streams")
    ap.add_argument("--check-gather", dest="check_gather", action="store_true", default=None,
                    help="N>1 (default ON there): after the timed steps, check on every rank that the gathered tensor holds "
                         "each rank's own last record (second route: all_gather_object of host copies) -> gather_verified")
    ap.add_argument("--no-check-gather", dest="check_gather", action="store_false")
    ap.add_argument("--graph", action="store_true",
                    help="every lane replays its batch from a hipGraph (one host call per batch instead of ~90 launches): the "
                         "per-rank host budget when eight ranks share one host; same bits")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false", default=True,
                    help="skip the two extra configurations measured behind the headline in the same process "
                         "(configs[2]: bf16 encoder, B=256; configs[3]'s per-GPU slice: fp32, B=256) -> `other_configs`")
    ap.add_argument("--force-exchange", action="store_true",
                    help="one rank only: initialise the process group anyway (RCCL with world size 1) and run the per-step "
                         "record exchange as the N>1 path does -- comm stream, ring of records, all_gather_into_tensor, "
                         "release_after -- so that the RCCL branch executes on a single MI355X (comm_ms_per_step, gather_verified)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="K-step regions timed in all (the first is `value`; all of them give value_spread)")
    args = ap.parse_args()

    if args.check_gather is None:
        args.check_gather = int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import torch.distributed as dist
    dist_on = world > 1 or args.force_exchange
    if args.force_exchange and world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.check_gather is False and "--no-check-gather" not in sys.argv:
            args.check_gather = True
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    if dist_on:
        from poserisk_release_amd import pipeline as pl
        pl.init_distributed(args.backend, dev)      # "nccl" = RCCL bound to this rank's device

    from poserisk_release_amd import pipeline as pl
    from poserisk_release_amd import synth
    from poserisk_release_amd.hmr import HMR
    from poserisk_release_amd.smpl_layer import SMPLLayer

    B = args.batch
    sd = synth.hmr_state_dict(seed=1)
    sm = synth.smpl_model(V=6890, seed=2)
    info = synth.EXAMPLE_INFO
    model = HMR(max_batch=B, precision=args.precision).to(dev)
    model.load_state_dict(sd)
    if args.streams > 0:
        model.set_streams(args.streams)
    layer = SMPLLayer(sm, device=dev, max_batch=max(B, 16))
    pipe = pl.FramePipeline(model, layer, info, with_verts=True, lanes=args.lanes, graph=args.graph)
    pipe.prepare(B, dev)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    crops = torch.rand((B, 3, 224, 224), generator=gen, device=dev, dtype=torch.float32)

    # the one exchange of the path: per-frame SMPL params, all-gathered on a side stream, off the critical path
    exchange = pl.RecordExchange(world, B, dev, n_buffers=max(args.lanes, 1)) if dist_on else None
    comm_stream = exchange.stream if exchange else None
    gathered = exchange.gathered if exchange else None
    comm_events = []                # (start, end) on the comm stream, one pair per timed step

    def step(timed=False):
        out = pipe(crops)            # asynchronous: this batch runs on its lane's stream
        if exchange is not None:
            if timed:
                pair = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
                exchange.step(out, on_stream=lambda st, i: pair[i].record(st))
                comm_events.append(tuple(pair))
            else:
                exchange.step(out)
        return out

    def fence():
        pipe.synchronize()
        if dist_on:
            comm_stream.synchronize()
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    def timed_region(timed_comm=False):
        """EXACTLY K steps between fences; returns (max over ranks, every rank's own seconds)."""
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=timed_comm)
        fence()
        mine = time.perf_counter() - t0
        if world == 1:
            return mine, [mine]
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = mine
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in t.cpu()]
        return max(per_rank), per_rank

    elapsed, per_rank_s = timed_region(timed_comm=True)      # THE measurement: `value`, `ms_per_step`
    comm_ms_per_step = None
    if dist_on and comm_events:
        torch.cuda.synchronize(dev)
        comm_ms_per_step = sum(a.elapsed_time(b) for a, b in comm_events) / len(comm_events)
    # the same region again: how much a K-step region moves from one repeat to the next (box noise, clocks)
    region_values = [args.steps * B * world / elapsed]
    for _ in range(max(args.repeats, 1) - 1):
        e, _unused = timed_region()
        region_values.append(args.steps * B * world / e)

    # How the batches in flight overlap on the GPU, from HIP events on each lane's own stream (a kernel trace under
    # rocprofv3 cannot show it for the fp32 step: tracing ~90 short launches per batch makes the host the limit).
    lanes_overlap = None
    if world == 1 and args.lanes > 1:
        base = torch.cuda.Event(enable_timing=True)
        fence()
        base.record(torch.cuda.current_stream(dev))
        pipe.record_times = []
        for _ in range(args.steps):
            step()
        fence()
        iv = sorted((base.elapsed_time(a), base.elapsed_time(b)) for a, b in pipe.record_times)
        pipe.record_times = None
        span = max(b for _, b in iv) - min(a for a, _ in iv)
        pts = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        depth, last, busy, multi = 0, pts[0][0], 0.0, 0.0
        for t, d in pts:
            if depth >= 1:
                busy += t - last
            if depth >= 2:
                multi += t - last
            depth += d
            last = t
        dur = [b - a for a, b in iv]
        lanes_overlap = {"batches": len(iv), "batch_ms_mean": round(sum(dur) / len(dur), 4),
                         "ms_per_step": round(span / len(iv), 4),
                         "batches_in_flight_mean": round(sum(dur) / span, 3),
                         "some_batch_running_frac": round(busy / span, 4),
                         "two_or_more_batches_running_frac": round(multi / span, 4),
                         "note": "hipEvent brackets on each lane's stream around pr_frames_forward, one more K-step region "
                                 "after the timed ones: a batch takes batch_ms_mean from its first to its last kernel while "
                                 "a new one completes every ms_per_step"}

    gather_verified = None
    if dist_on and args.check_gather:
        # one more step, fenced, then every rank's record by a second route; rows [r*B, (r+1)*B) must be rank r's
        last_out = step()
        fence()
        mine = exchange.last_record().cpu()
        parts = [None] * world
        dist.all_gather_object(parts, mine)                  # second route: pickled host copies
        got = gathered.cpu()
        ok = all(torch.equal(got[r * B:(r + 1) * B], parts[r]) for r in range(world))
        ok = ok and not any(torch.equal(parts[0], parts[r]) for r in range(1, world))   # ranks see different crops
        ok = ok and torch.equal(got, pl.pack_record(last_out).cpu()) if world == 1 else ok   # one rank: the record itself
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_verified = bool(flag.item())

    # Transparency: the same K steps with ONE batch in flight (each step waits for the previous one on the
    # same stream), so the gain from overlapping whole batches is visible next to `value`.
    serial_fps = None
    if world == 1 and args.lanes > 1:
        one = pl.FramePipeline(model, layer, info, with_verts=True, lanes=1)
        for _ in range(3):
            one(crops)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one(crops)
        torch.cuda.synchronize(dev)
        serial_fps = args.steps * B / (time.perf_counter() - t1)

    roofline = None
    if rank == 0 and not args.no_roofline:
        # Same K steps again with every conv launch bracketed by hipEvents on the launch stream
        # (kept out of the timed region so the events do not perturb `value`; in this pass one batch
        # is in flight at a time so that each bracket times one kernel alone).
        prof = pl.FramePipeline(model, layer, info, with_verts=True, lanes=1)
        model.profile_enable(True)
        for _ in range(args.steps):
            prof(crops)
        torch.cuda.synchronize(dev)
        ms, cnt, flops_per_frame, mfma_flops_per_frame = model.profile_read(with_mfma_flops=True)
        model.profile_enable(False)
        plan_launches, plan_wino = model.plan_counts(B)
        if int(cnt.sum()) != plan_launches * args.steps:
            raise SystemExit(f"bench.py: {int(cnt.sum())} conv event brackets in {args.steps} steps, the plan says "
                             f"{plan_launches} per step")
        # algorithmic (direct-convolution) FLOP of the K steps: SURVEY.md 8d's 8.174 GFLOP per frame, whatever
        # form a layer is computed in
        total_flop = float(flops_per_frame.sum()) * B * args.steps
        achieved = total_flop / (float(ms.sum()) * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        traffic, traffic_note = None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_hbm_traffic_" +
                                               ("b64" if args.precision == "fp32" else "b256_bf16") + ".json")))
        tpath = tfiles[-1] if tfiles else ""
        if ((B == 64 and args.precision == "fp32") or (B == 256 and args.precision == "bf16")) and tpath:
            # PMC counters need their own rocprofv3 passes (FETCH_SIZE, WRITE_SIZE), so the per-launch HBM
            # bytes come from the committed summary of those passes over this same workload.
            tj = json.load(open(tpath))
            traffic = tj["conv_hbm_bytes_per_launch"]
            traffic_note = ("HBM bytes per launch of the conv family (" + os.path.basename(tpath) + "; per conv layer: " + str(tj.get("conv_hbm_bytes_per_layer")) + "), " + tj["source"] + "; " + tj["correction"])
        executed = float(mfma_flops_per_frame.sum()) * B * args.steps / (float(ms.sum()) * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": ("conv_dma_f32 + conv1x1_regw_f32 + conv3x3_conv1x1_f32 + stem_pool_f32 (the 53 conv layers of a step in 47 launches: "
                                                "the stem with its max-pool is one kernel, a downsample branch rides in its conv3's K loop, layer1's "
                                                "conv2+conv3 pairs are one kernel, the 1x1 layers with K <= 256 keep their weights in registers; 10 layers "
                                                "in Winograd form -- F(4x4,3x3) on the points 0, +-11/16, +-3/2 -- = transform + 36 grouped GEMMs + "
                                                "transform, timed as one)"
                                                if args.precision == "fp32" else
                                                "conv_dma_bf16 + conv_bal_bf16 + bottleneck64_bf16 + bottleneck128_bf16 + bottleneck256_bf16 + "
                                                "stem_pool_bf16 + expand_res_bf16 (53 conv layers in 27 launches per step at B=256, 37 at batches that "
                                                "do not fill the CUs: the stem with its max-pool is one kernel, each of layer1's three blocks and of "
                                                "layer2's three plain blocks one persistent kernel, each of layer3's five plain blocks one kernel with a "
                                                "frame per workgroup, layer2's first expansion keeps its weights in registers, the other downsample "
                                                "branches ride in their conv3's K loop; the remaining 1x1 / 3x3 layers of layer2.0, layer3.0 and layer4 "
                                                "run on the evenly dealt persistent kernel where it pays)"),
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4),
                    "achieved_is": "ALGORITHMIC direct-convolution FLOP (SURVEY.md 8d: 8.174 GFLOP per frame) / measured conv time; "
                                   "not the matrix pipes' utilisation",
                    "mfma_executed_tflops": round(executed, 2),
                    "mfma_executed_frac": round(executed / peak, 4),
                    "mfma_executed_is": "FLOP the matrix pipes execute (K padding included, 36 products per 4x4 Winograd tile) / "
                                        "the same time; the PMC counter SQ_VALU_MFMA_BUSY_CYCLES of the profiled run is in "
                                        "profiles/*_pmc_mfma_busy_b64.txt",
                    "traffic": traffic,
                    "traffic_note": traffic_note,
                    # what the library says a step launches; scripts/pmc_summary.py checks its counter passes against it
                    "conv_launches_per_step": int(cnt.sum()) // args.steps,
                    "winograd_layers": plan_wino,
                    "conv_kernels_per_step": plan_launches + 2 * plan_wino,
                    "avg_launch_us": round(float(ms.sum()) / max(int(cnt.sum()), 1) * 1e3, 2),
                    "flop_per_launch": round(total_flop / max(int(cnt.sum()), 1), 1),
                    "conv_ms_per_step": round(float(ms.sum()) / args.steps, 4),
                    # this pass runs ONE batch in flight (each bracket must time one kernel alone); the headline runs
                    # `config.batches_in_flight` of them, so conv_ms_per_step belongs beside ms_per_step_same_mode, not
                    # beside the line's ms_per_step
                    "batches_in_flight": 1,
                    "ms_per_step_same_mode": round(B / serial_fps * 1e3, 4) if serial_fps else round(elapsed / args.steps * 1e3, 4)}
    smpl_lbs = None
    if rank == 0 and not args.no_roofline:
        # SURVEY.md 8d also asks for the SMPL forward's achieved HBM rate: flags + pose + skin kernels of one
        # batch (mesh + joints), timed with events on the stream they are launched on (torch's current one).
        def smpl_rate(lay, n):
            pose = torch.from_numpy(synth.poses(n, seed=1)).to(dev)
            betas = torch.from_numpy(synth.betas(n, seed=2)).to(dev)
            for _ in range(5):
                lay(pose, betas)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                lay(pose, betas)
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) / 50 * 1e3
            nbytes = SMPL_CONST_BYTES + n * SMPL_BYTES_PER_FRAME
            return us, nbytes

        def smpl_roofline(us, nbytes, n):
            # the bound is whichever roof is lower at this batch: 2 x 7.74 MFLOP per frame against 19.35 MB of model
            # constants per launch + 83 KB per frame is 40 FLOP per algorithmic byte at 64 frames and 97 at 256, the machine
            # balance (157.3 TFLOP/s of packed-fp32 VALU -- the kernels use no MFMA -- over 8 TB/s) is 19.7: VALU-bound from 28 frames up
            tflops, gbps = SMPL_FLOP_PER_FRAME * n / us / 1e6, nbytes / us / 1e3
            valu = SMPL_FLOP_PER_FRAME * n / nbytes > PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)
            o = {"bound": "valu" if valu else "hbm",
                 "achieved": round(tflops if valu else gbps, 2), "peak": PEAK_F32_MFMA_TFLOPS if valu else PEAK_HBM_GBPS,
                 "unit": "TFLOP/s" if valu else "GB/s",
                 "frac": round(tflops / PEAK_F32_MFMA_TFLOPS if valu else gbps / PEAK_HBM_GBPS, 4),
                 "us_per_forward": round(us, 2), "frames": n, "bytes_per_forward": nbytes,
                 "flop_per_byte": round(SMPL_FLOP_PER_FRAME * n / nbytes, 1),
                 "achieved_gbps": round(gbps, 1), "hbm_frac": round(gbps / PEAK_HBM_GBPS, 4),
                 "achieved_tflops": round(tflops, 2), "valu_frac": round(tflops / PEAK_F32_MFMA_TFLOPS, 4)}
            return o

        us, nbytes = smpl_rate(layer, B)
        smpl_lbs = smpl_roofline(us, nbytes, B)
        smpl_lbs["note"] = ("packed-fp32 VALU streaming over LDS-staged coefficients, no MFMA (north star); both roofs reported, "
                            "`bound` names the lower one at this batch (SURVEY 8d priced it against HBM only)")
        if B != 256:
            big = SMPLLayer(sm, device=dev, max_batch=256)    # handles of more than 128 frames use smpl_skin_rows
            us2, nb2 = smpl_rate(big, 256)
            smpl_lbs["at_b256"] = smpl_roofline(us2, nb2, 256)
            del big
    other_configs = None
    if rank == 0 and world == 1 and args.other_configs and args.precision == "fp32" and B == 64:
        # configs[2] and configs[3]'s per-GPU slice, driver-observed: same process, fresh handles (the headline's are
        # released first), the driver's own --steps / --warmup
        del pipe
        model._release()
        torch.cuda.empty_cache()
        other_configs = [
            measure_other_config("bf16", 256, 2, args.steps, args.warmup, dev, sd, sm, info,
                                 "configs[2]: batch=256 bf16 encoder (CDNA4 bf16 MFMA), fp32 SMPL LBS, 1 GPU"),
            measure_other_config("fp32", 256, 2, args.steps, args.warmup, dev, sd, sm, info,
                                 "configs[3]'s per-GPU slice: batch=256 (2048 frames over 8 GPUs), ResNet-50+SMPL fp32, 1 GPU"),
        ]
    dist_info = None
    if dist_on:
        # did the backend see N ranks on N devices?  answered by the record itself
        me = {"rank": rank, "device": str(dev), "device_name": torch.cuda.get_device_name(dev),
              "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)}
        seen = [None] * world
        dist.all_gather_object(seen, me)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001  (a build without the binding still has to print its line)
            ver = f"unavailable ({type(e).__name__})"
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "rccl_version (torch.cuda.nccl.version)": ver, "ranks": seen,
                     "distinct_devices": len({(d["device"], d["pci_bus_id"]) for d in seen})}
        dist.barrier()

    if rank == 0:
        frames = args.steps * B * world
        value = frames / elapsed
        line = {"metric": "frames/sec (224x224 crops) through SPIN ResNet-50 + regressor + SMPL LBS + REBA/RULA",
                "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if args.precision == "fp32" else "bf16 (encoder; f32 accumulate, regressor/SMPL f32)",
                "data": "synthetic (uniform [0,1) crops, seeded random-init SPIN weights and SMPL model)",
                "config": {"workload": ("configs[1]: batch=64 random 224x224 crops per GPU, ResNet-50+SMPL fp32"
                                        if args.precision == "fp32" else
                                        f"configs[2]: batch={B} bf16 encoder (CDNA4 bf16 MFMA), fp32 SMPL LBS"),
                           "frames_per_gpu_per_step": B, "global_batch": B * world,
                           "batches_in_flight": args.lanes, "hipgraph_replay": bool(args.graph),
                           "exchange": "all-gather of 916-B per-frame SMPL params per step" if dist_on else "none",
                           "dist_backend": dist.get_backend() if dist_on else None,
                           "dist_world_size": dist.get_world_size() if dist_on else 1},
                "conv_roofline_frames_per_s_per_gpu": round(PEAK_F32_MFMA_TFLOPS * 1e3 / CONV_GFLOP_PER_FRAME, 1),
                "frac_of_conv_roofline": round(value / world / (PEAK_F32_MFMA_TFLOPS * 1e3 / CONV_GFLOP_PER_FRAME), 4)}
        if args.precision != "fp32":
            line["conv_roofline_frames_per_s_per_gpu"] = round(PEAK_BF16_MFMA_TFLOPS * 1e3 / CONV_GFLOP_PER_FRAME, 1)
            line["frac_of_conv_roofline"] = round(value / world / line["conv_roofline_frames_per_s_per_gpu"], 4)
        rv = sorted(region_values)
        line["value_spread"] = {"regions": len(rv), "min": round(rv[0], 1), "median": round(rv[len(rv) // 2], 1),
                                "max": round(rv[-1], 1),
                                "note": "frames/s of each timed K-step region; `value` is the first one"}
        if dist_on:
            ms_rank = [t / args.steps * 1e3 for t in per_rank_s]
            line["per_rank_ms_per_step"] = {"min": round(min(ms_rank), 4), "max": round(max(ms_rank), 4),
                                            "all": [round(v, 4) for v in ms_rank]}
            line["comm_ms_per_step"] = None if comm_ms_per_step is None else round(comm_ms_per_step, 4)
            line["comm_note"] = ("events on rank 0's side stream around pack + all_gather_into_tensor of one step "
                                 "(overlaps the next batch; not on the critical path unless it exceeds ms_per_step)")
            line["dist"] = dist_info
        if gather_verified is not None:
            line["gather_verified"] = gather_verified
        if serial_fps is not None:
            line["frames_per_s_one_batch_in_flight"] = round(serial_fps, 1)
        if lanes_overlap is not None:
            line["lanes_overlap"] = lanes_overlap
        if roofline is not None:
            line["roofline"] = roofline
        if smpl_lbs is not None:
            line["smpl_lbs"] = smpl_lbs
        if other_configs is not None:
            line["other_configs"] = other_configs
        from poserisk_release_amd import _lib
        line["library"] = {"path": os.path.relpath(_lib.LIB_PATH, REPO), "build": _lib.load().pr_build_info().decode(),
                           "abi": _lib.load().pr_abi_version()}
        if world == 1 and args.cpu_frames > 0:
            line["cpu_baseline"] = cpu_baseline(sd, sm, info, args.cpu_frames)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
