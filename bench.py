#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

One STEP = one whole frame of the hot path: the reference's default scene
(src/Object.hs:147-148) at 1920x1080, 1024 samples per pixel, maxBounces 8
(BASELINE.json metric; configs[1]/[2]), with inputs (scene, canvas) already
resident in HBM:
    reset canvas -> pt_trace(spp / N on each of N GPUs, disjoint global sample
    indices) -> [N > 1: ONE RCCL all-reduce (sum) of the fp32 canvas over xGMI]
    -> fused RGB8 output kernel on rank 0.
The total work per step is fixed as N grows, so "scaling" is "strong".
Frames are double-buffered (pathtracer_amd.distributed.PipelinedFrames), for
every N alike: frame k's all-reduce and output run on their own streams while
frame k+1 is traced into the other canvas on a second trace stream; the timed
region ends with a full drain, so all K frames are complete (traced, reduced,
quantised) in it.  --no-pipeline runs the steps strictly in line.

value = closest-hit searches executed by all ranks in the timed steps / wall
time (Mrays/s, a ray = one closestObjectTo, src/Lib.hs:81).  The searches are
counted by the kernel itself; the per-pixel primary search is executed once per
launch, not once per sample (the primary ray has no jitter, src/Lib.hs:41-51),
and only executed searches are counted.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
        (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (vector)"


def flops_per_segment(objs):
    """Algorithmic flops of one segment, SURVEY.md 8(d): 12 per plane test, 26
    per disk test, + scatter (35 diffuse / 70 specular; sample-weighted by the
    share of scattering objects of each kind)."""
    from pathtracer_amd import Disk
    f = sum(26 if isinstance(o.form, Disk) else 12 for o in objs)
    scat = [70 if o.reflective else 35 for o in objs if not o.emittance > 0]
    return f + (sum(scat) / len(scat) if scat else 0)


def cpu_baseline(height, width, depth, bounces, threads, target_s=15.0):
    """The oracle (CPU fp64 restatement of the reference, StdGen stream) timed
    on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    p = O.params(height, width, depth, bounces, O.STDGEN)
    tr = O.Tracer(p, O.default_scene(), threads=threads)
    t0 = time.perf_counter()
    tr.trace(1)
    dt1 = time.perf_counter() - t0
    spp = 1
    more = int(max(0, min(512, target_s / max(dt1, 1e-3) - 1)))
    if more:
        tr.trace(more)
        spp += more
    dt = time.perf_counter() - t0
    return {"value": round(tr.segments.value / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads,
            "kind": "port",
            "sample": f"default scene {height}x{width}, {spp} spp, depth {bounces}: "
                      f"{tr.segments.value} segments in {dt:.2f} s; fp64 C restatement of the reference "
                      f"(oracle/pt_oracle.c, StdGen stream, OpenMP over rows), not the Haskell program"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1920, help="Params.height = image width in pixels (F10)")
    ap.add_argument("--width", type=int, default=1080, help="Params.width = image height in pixels (F10)")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--camera-depth", type=float, default=500.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-primary-reuse", action="store_true")
    ap.add_argument("--no-steal", action="store_true", help="PT_FLAG_NO_STEAL: strict per-lane sample order")
    ap.add_argument("--scene", default="default",
                    help="default | disks1000 | spheres1000 | disks10000 | spheres10000: the synthetic scenes of "
                         "BASELINE.json configs[3]/[4] (SURVEY.md 8d generator); pass their --height/--width/--spp/--bounces")
    ap.add_argument("--no-bvh", action="store_true", help="PT_FLAG_NO_BVH: linear scan over all objects")
    ap.add_argument("--no-specialise", action="store_true",
                    help="keep the general trace kernel: no pt_scene_specialise (kernel compiled for the scene) at set-up")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="trace, all-reduce and output strictly in line instead of overlapped with the next frame")
    args = ap.parse_args()
    # stdout carries exactly ONE line, the JSON: anything a library prints to file descriptor 1
    # (RCCL's start-up banner does) goes to stderr instead
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import pathtracer_amd as pa
    from pathtracer_amd.distributed import PipelinedFrames, allreduce_canvas, canvas_tensor, shard_samples

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libpt has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    # PT_BENCH_FORCE_DIST=1 drives the RCCL all-reduce path with a world of 1 (rehearsal on a 1-GPU box)
    if world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    my_spp, s_off, s_stride = shard_samples(args.spp, rank, world)
    P = pa.Params(args.height, args.width, args.camera_depth, args.spp, args.bounces)
    flags = (pa.PT_FLAG_NO_PRIMARY_REUSE if args.no_primary_reuse else 0) | (pa.PT_FLAG_NO_STEAL if args.no_steal else 0) \
        | (pa.PT_FLAG_NO_BVH if args.no_bvh else 0)
    if args.scene == "default":
        scene, scene_name = pa.objects, "reference default scene (4 objects)"
    else:
        kind, count = ("disks", args.scene[5:]) if args.scene.startswith("disks") else ("spheres", args.scene[7:])
        n_obj = int(count)
        gen = pa.random_disk_scene if kind == "disks" else pa.random_sphere_scene
        scene = gen(n_obj, 1 if n_obj <= 1000 else 2)
        scene_name = (f"synthetic {n_obj}-object random-{kind[:-1]} scene (SURVEY.md 8d generator, seed "
                      f"{1 if n_obj <= 1000 else 2}" + ("; spheres are an extension without reference parity" if kind == "spheres" else "") + ")")
    # An explicit stream: libpt cannot be pointed at the legacy null stream, and
    # the all-reduce must be ordered after the trace kernel on the SAME stream.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    pipelined = not args.no_pipeline         # the same frame loop for every N; N = 1 simply has no all-reduce
    rs = [pa.Renderer(P, scene, seed=0, device=local_rank, sample_offset=s_off, sample_stride=s_stride,
                      flags=flags) for _ in range(2 if pipelined else 1)]
    # set-up, like the scene upload: the trace kernel compiled for this scene (pt_scene_specialise; by-value
    # scenes on the default flags only -- elsewhere, or without hiprtc on the machine, the general kernel stays)
    specialised = False
    if not args.no_specialise and len(scene) <= 8 and not (args.no_steal or args.no_primary_reuse):
        specialised = all([q.specialise() for q in rs])
        if not specialised:
            print("bench.py: pt_scene_specialise unavailable: " + rs[0].last_error(), file=sys.stderr)
    for q in rs:
        q.set_stream(stream.cuda_stream)      # kernels, all-reduce and events share this stream
    r = rs[0]

    # the contexts' canvases as torch tensors (no copy), for the all-reduce
    canvases = [canvas_tensor(q, local_rank) for q in rs]
    canvas = canvases[0]
    rgbs = [torch.empty(args.height * args.width * 3, dtype=torch.uint8, device="cuda") for _ in rs]
    rgb = rgbs[0]
    pipe = PipelinedFrames(rs, canvases, rgbs, rank=rank, spp_total=args.spp, my_spp=my_spp,
                           mode=pa.PT_RGB8_MEAN_CLAMP, dist=dist) if pipelined else None

    def step():
        if pipe is not None:
            pipe.step()
            return
        r.reset()
        r.trace(my_spp)
        if dist is not None:
            allreduce_canvas(canvas)
        if rank == 0:
            r.rgb8_device(rgb.data_ptr(), pa.PT_RGB8_MEAN_CLAMP, args.spp)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # at least one untimed step per context (two when frames are double-buffered): a context's first launch
    # allocates its slice plane and computes and keeps the primary hits, which is set-up, not a step
    for _ in range(max(args.warmup, len(rs))):
        step()
    fence()
    if pipe is not None:
        r = rs[(pipe.k - 1) & 1]               # the context of the last frame: its counter holds one step
    segs_step = torch.tensor([r.segments], dtype=torch.float64, device="cuda")

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0

    # a separate, individually event-timed pass for the roofline average (HIP events on the launch stream)
    r.set_timing(True)
    ev = []
    for _ in range(min(args.steps, 5)):
        r.reset()
        r.trace(my_spp)
        ms_one, launches = r.last_trace_ms()
        ev.append(ms_one)
    fence()
    kern_ms = sum(ev) / len(ev) / max(launches, 1)

    t_el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(segs_step, op=dist.ReduceOp.SUM)
    elapsed = float(t_el.item())
    total_segs_step = float(segs_step.item())

    if rank == 0:
        value = total_segs_step * args.steps / elapsed / 1e6
        # algorithmic flops: SURVEY 8(d)'s per-segment figure for a scene searched by the linear scan; a
        # scene searched through the spatial index executes far fewer object tests than that formula
        # counts, so no roofline fraction is claimed for it
        indexed = len(scene) > 8 and not args.no_bvh
        F = flops_per_segment(scene)
        segs_launch_rank0 = r.segments / max(launches, 1)
        achieved = F * segs_launch_rank0 / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("spp") == args.spp and tj.get("height") == args.height:
                    traffic = tj.get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (samples x bounces) at 1080p 1024spp",
            "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{scene_name}, Params.height={args.height} "
                                   f"Params.width={args.width} ({args.height}x{args.width} px image), "
                                   f"{args.spp} spp, maxBounces {args.bounces}, cameraDepth {args.camera_depth:g}; "
                                   f"sample-split over {world} GPU(s) + one RCCL all-reduce + RGB8 output",
                       "segments_per_step": int(total_segs_step),
                       "paths_per_step": args.height * args.width * args.spp,
                       "primary_reuse": not args.no_primary_reuse,
                       "scene_kernel": "compiled for this scene at set-up (pt_scene_specialise)" if specialised
                                       else "general",

                       "parallelism": f"sample-split x{world}",
                       "frame_pipeline": "double-buffered: all-reduce + output of frame k overlap the trace of "
                                         "frame k+1" if pipe is not None else "none"},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                         "traffic": traffic,
                         "kernel": "trace_kernel" if (args.no_steal or args.no_primary_reuse) else
                                   ("trace_static_kernel" if len(scene) <= 8 else
                                    "trace_bvh_kernel" if indexed else "trace_steal_kernel"),
                         "kernel_ms": round(kern_ms, 4),
                         "flops_per_segment": F, "segments_per_launch": int(segs_launch_rank0),
                         "note": "fp32 VALU-bound path (no MFMA, ~0 HBM bytes per segment); peak = fp32 vector "
                                 "157.3 TFLOP/s (= the fp32 MFMA peak)"},
        }
        if indexed:
            out["roofline"].update(achieved=None, frac=None, traffic=None,
                                   note="searched through the BVH: the linear-scan flop formula does not apply; "
                                        "see value (Mrays/s) and DESIGN.md 3.4")
        if args.scene != "default":
            out["metric"] = "Mrays/s (samples x bounces)"
        if world == 1 and not args.no_cpu_baseline and args.scene == "default":
            threads = args.cpu_threads or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(args.height, args.width, args.camera_depth, args.bounces, threads)
        print(json.dumps(out), file=json_out, flush=True)

    for q in rs:
        q.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
