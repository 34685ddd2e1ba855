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
N > 1 runs one rank process per GPU.  Under torch.distributed.run the ranks
exist already (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment);
typed as plain `python bench.py --gpus N` this process -- before it imports
torch or touches a GPU -- starts the N rank processes itself (self_launch) and
relays rank 0's JSON line.  The frame's ONE collective is libpt's own
ncclAllReduce (pt_comm_init_rank / pt_allreduce_sum, include/pt.h: RCCL behind
the C ABI); torch.distributed (gloo, CPU) only carries the RCCL id, the barriers
and the max-over-ranks of the time.  --collective torch times torch's RCCL
all-reduce on the same canvases instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (vector)"
GATHER_PEAK_GBS = 15980.0          # fully divergent 16-byte gathers from an L1-resident (16-KB) table, measured: 999 G lane-gathers/s,
                                   # tools/ubench_gather.hip, profiles/r03_bvh_walk.txt (round 2's 8,216 came from a 64-KB table, which
                                   # does not fit the 32-KB L1; an L2-resident table gives 280 G distinct records/s = 4.5 TB/s)


def flops_per_segment(objs):
    """Algorithmic flops of one segment, SURVEY.md 8(d): 12 per plane test, 26
    per disk test, + scatter (35 diffuse / 70 specular; sample-weighted by the
    share of scattering objects of each kind)."""
    from pathtracer_amd import Disk
    f = sum(26 if isinstance(o.form, Disk) else 12 for o in objs)
    scat = [70 if o.reflective else 35 for o in objs if not o.emittance > 0]
    return f + (sum(scat) / len(scat) if scat else 0)


def _time_oracle(height, width, depth, bounces, threads, target_s):
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
    return tr.segments.value, spp, dt


def cpu_baseline(height, width, depth, bounces, threads, target_s=12.0):
    """The oracle (CPU fp64 restatement of the reference, StdGen stream) timed
    on this box's host cores on a bounded sample of the same workload: with
    `threads` OpenMP threads, and with one (the reference itself
    never runs in parallel, SURVEY F11, so the single-thread figure is the one
    closest to what the Haskell program does)."""
    segs, spp, dt = _time_oracle(height, width, depth, bounces, threads, target_s)
    segs1, spp1, dt1 = _time_oracle(height, width, depth, bounces, 1, target_s / 2)
    what = "fp64 C restatement of the reference (oracle/pt_oracle.c, StdGen stream), not the Haskell program"
    return {"value": round(segs / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads,
            "kind": "port",
            "sample": f"default scene {height}x{width}, {spp} spp, depth {bounces}: "
                      f"{segs} segments in {dt:.2f} s; {what}; OpenMP over rows",
            "single_thread": {"value": round(segs1 / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                              "sample": f"the same frame, {spp1} spp: {segs1} segments in {dt1:.2f} s"}}


def rank_environments(n, port, base=None):
    """The environment of each of `n` rank processes of one node, as
    torch.distributed.run would set it (what main() reads)."""
    base = dict(os.environ if base is None else base)
    envs = []
    for r in range(n):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        envs.append(e)
    return envs


def visible_devices_from_env(env=None):
    """How many GPUs the environment lets this process see, or None if it does not say: a launcher may hand each rank
    its own HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is honoured by HIP too); device ordinals
    then count the VISIBLE devices.  An empty value hides every device."""
    env = os.environ if env is None else env
    n = None
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if key in env:
            k = len([x for x in env[key].split(",") if x.strip() != ""])
            n = k if n is None else min(n, k)
    return n


def device_for_rank(local_rank, n_visible):
    """The HIP ordinal of a rank's GPU among the devices its process can see: its local rank where the process sees
    the whole node, ordinal 0 where the launcher gave it exactly one device, round-robin in between."""
    if n_visible is None or n_visible <= 0:
        raise RuntimeError("bench.py: no visible GPU for this rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES hide them all?)")
    return local_rank if local_rank < n_visible else local_rank % n_visible


def agree_on_libpt_collective(dist, rank, world, make_id, init_rank, own_world_of_one=False):
    """Set up libpt's RCCL communicator on every rank, or agree -- all ranks together -- that it cannot be done.
    Every rank executes the SAME sequence of control-plane (gloo) collectives whatever fails where: rank 0's id, or
    None if it could not be made, is ALWAYS broadcast; every rank then tries to join and the outcome is a MIN
    all-reduce.  (Round 2 made the id inside the try that also held the broadcast: a failure on rank 0 skipped the
    broadcast the other ranks were waiting in.)  own_world_of_one: every rank makes an id of its own and a
    communicator of one rank (the rehearsal on a one-GPU box); the agreement still runs.
    Returns (ok, reason)."""
    import torch
    uid, why = None, ""
    if own_world_of_one or rank == 0:
        try:
            uid = make_id()
        except Exception as e:                          # noqa: BLE001 (reported, then decided by all ranks together)
            why = f"rank {rank}: no RCCL id through libpt: {e}"
    if not own_world_of_one:
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
        if uid is None and not why:
            why = "rank 0 could not make an RCCL id through libpt"
    ok = 0
    if uid is not None:
        try:
            init_rank(uid, 0 if own_world_of_one else rank, 1 if own_world_of_one else world)
            ok = 1
        except Exception as e:                          # noqa: BLE001
            why = f"rank {rank}: pt_comm_init_rank: {e}"
    if why:
        print("bench.py: libpt collective unavailable: " + why, file=sys.stderr, flush=True)
    flag = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1, why


def self_launch(n, argv):
    """`python bench.py --gpus N` typed as is: start the N rank processes (this
    same script with the rank environment), relay rank 0's stdout -- the ONE JSON
    line -- and exit with the worst exit status.  Called before torch is imported
    or a GPU touched: nothing that initialised a GPU is forked or replaced.  If a
    rank dies the others are ended (by their own PIDs), not left in a rendezvous."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r, env in enumerate(rank_environments(n, port)):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0:
                rc = rc or code
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    out = procs[0].stdout.read().decode(errors="replace")
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


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
    ap.add_argument("--pipeline-depth", type=int, default=2, choices=(2, 3),
                    help="frames in flight (canvases and trace streams).  3 measured 1-4 %% slower than 2 on one GPU with "
                         "libpt's all-reduce over a world of 1 in the loop (tools/pipeline_depth_ab.sh, DESIGN.md 6)")
    ap.add_argument("--comm-shared", action="store_true",
                    help="N > 1, libpt collective: one RCCL communicator per rank shared by its pipelined contexts "
                         "(pt_comm_share) instead of one per context.  Measured on one GPU (RCCL world of 1, a rank's "
                         "128-spp share): 1.54 against 1.45 ms per step -- RCCL orders the launches of one communicator "
                         "across streams, which ties frame k+1's all-reduce to frame k's (profiles/r03_frame_pipeline.txt)")
    ap.add_argument("--gate", action="store_true",
                    help="a frame's trace starts when the frame before it has dispatched its last workgroup (pt_trace_gate) "
                         "instead of whenever its canvas is free; measured to change the overlap, not the throughput")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="trace, all-reduce and output strictly in line instead of overlapped with the next frame")
    ap.add_argument("--watchdog-s", type=float, default=900.0,
                    help="end the process (exit status 124, a line on stderr) if it has not finished by then: a rank stuck in a "
                         "rendezvous or a collective must not hold its GPU until somebody else's limit; 0 = off")
    ap.add_argument("--collective", choices=("libpt", "torch"), default="libpt",
                    help="N > 1: libpt's own ncclAllReduce behind the C ABI (default) or torch.distributed's")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.watchdog_s > 0:
        import threading

        def expire():
            print(f"bench.py: rank {os.environ.get('RANK', '0')}: not finished after {args.watchdog_s:.0f} s, giving up "
                  "(--watchdog-s)", file=sys.stderr, flush=True)
            os._exit(124)
        dog = threading.Timer(args.watchdog_s, expire)
        dog.daemon = True
        dog.start()
    # stdout carries exactly ONE line, the JSON: anything a library prints to file descriptor 1
    # (RCCL's start-up banner does) goes to stderr instead
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if os.environ.get("PT_BENCH_DRY_RUN") == "1":
        # rank bookkeeping only (tests/test_bench_launcher.py, no GPU): rendezvous over gloo, count the ranks
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1, local_rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        out = {"dry_run": True, "n_gpus": world, "rank_sum": int(t[0]), "local_rank_sum": int(t[1])}
        # PT_BENCH_DRY_FAIL = "id" | "init<rank>": the libpt-collective agreement with stand-ins that fail there
        fail_at = os.environ.get("PT_BENCH_DRY_FAIL")
        if fail_at:
            def make_id():
                if fail_at == "id":
                    raise RuntimeError("forced: no id")
                return b"x" * 128

            def init_rank(uid, rk, n):
                if fail_at == f"init{rank}":
                    raise RuntimeError("forced: init")
            ok, _ = agree_on_libpt_collective(dist, rank, world, make_id, init_rank)
            t2 = torch.tensor([1], dtype=torch.int64)
            dist.all_reduce(t2)                          # the ranks are still in step after the agreement
            out.update(libpt_collective=ok, in_step=int(t2[0]) == world)
        if rank == 0:
            print(json.dumps(out), file=json_out, flush=True)
        dist.destroy_process_group()
        return

    import torch
    import pathtracer_amd as pa
    from pathtracer_amd.distributed import PipelinedFrames, allreduce_canvas, canvas_tensor, shard_samples

    if torch.cuda.device_count() == 0 or not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libpt has no CPU path")
    # PT_BENCH_REHEARSE_ONE_GPU=1: every rank on device 0, frames NOT combined across ranks (RCCL refuses two ranks of
    # one communicator on one device): a rehearsal of the N > 1 control flow -- launcher, gloo rendezvous, sample
    # split, fences, max of the time, sum of the segments -- on a one-GPU box; with --collective libpt every rank runs
    # libpt's all-reduce over a communicator of its own with one rank, so the collective call sits in the multi-rank
    # loop.  The line it prints says so and is not a measurement.
    rehearsal = os.environ.get("PT_BENCH_REHEARSE_ONE_GPU") == "1" and world > 1
    # the rank's GPU among the devices this process can see (a launcher may have set HIP_VISIBLE_DEVICES /
    # ROCR_VISIBLE_DEVICES per rank; torch.cuda.device_count() counts the visible ones without touching a GPU)
    local_rank = 0 if rehearsal else device_for_rank(local_rank, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    # PT_BENCH_FORCE_DIST=1 drives the collective path with a world of 1 (rehearsal on a 1-GPU box)
    multi = world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1"
    if multi:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: RCCL's and gloo's bootstrap sockets on the loopback interface, whatever else the box has
            # (data moves over xGMI / shared memory either way); an interface named by the caller wins
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")          # control plane only: the RCCL id, barriers, max of the time

    my_spp, s_off, s_stride = shard_samples(args.spp, rank, world)
    P = pa.Params(args.height, args.width, args.camera_depth, args.spp, args.bounces)
    flags = (pa.PT_FLAG_NO_PRIMARY_REUSE if args.no_primary_reuse else 0) | (pa.PT_FLAG_NO_STEAL if args.no_steal else 0) \
        | (pa.PT_FLAG_NO_BVH if args.no_bvh else 0)
    if args.scene == "default":
        scene, scene_name = pa.objects, "reference default scene (4 objects)"
    else:
        kind, count = ("disks", args.scene[5:]) if args.scene.startswith("disks") else ("spheres", args.scene[7:])
        n_obj = int(count)
        gen_seed = 1 if n_obj <= 1000 else 2
        scene = pa.synthetic_scene(n_obj, gen_seed, kind)      # libpt's SplitMix64 generator (pt_synthetic_scene)
        scene_name = (f"synthetic {n_obj}-object random-{kind[:-1]} scene (SURVEY.md 8d generator pt_synthetic_scene, "
                      f"SplitMix64 seed {gen_seed}" + ("; spheres are an extension without reference parity" if kind == "spheres" else "") + ")")
    # An explicit stream: libpt cannot be pointed at the legacy null stream, and
    # the all-reduce must be ordered after the trace kernel on the SAME stream.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    pipelined = not args.no_pipeline         # the same frame loop for every N; N = 1 simply has no all-reduce
    depth = args.pipeline_depth
    rs = [pa.Renderer(P, scene, seed=0, device=local_rank, sample_offset=s_off, sample_stride=s_stride,
                      flags=flags) for _ in range(depth if pipelined else 1)]
    # set-up, like the scene upload: the trace kernel compiled for this scene (pt_scene_specialise; by-value
    # scenes on the default flags only -- elsewhere, or without hiprtc on the machine, the general kernel stays)
    specialised = False
    if not args.no_specialise and len(scene) <= 16 and not (args.no_steal or args.no_primary_reuse):
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
    # The frame's collective.  libpt: one RCCL communicator per pipelined context (frames k and k+1 are in flight on
    # two streams), each joined with an id of rank 0's handed round over gloo; --comm-shared: ONE per rank, which
    # the other contexts share (pt_comm_share) -- fewer channels and proxies, but measured 6 % slower per step on
    # one GPU because RCCL then orders the two streams' launches.  If RCCL cannot be reached through libpt on this
    # machine every rank falls back to torch's all-reduce together, and the JSON line says which one ran.
    collective, comm, comm_per_ctx = None, {}, []
    if multi:
        collective = args.collective
        if rehearsal and collective != "libpt":
            collective = "none (rehearsal on one GPU: frames are NOT combined)"
        if collective == "libpt":
            def init_rank(uid, rk, n):
                rs[0].comm_init_rank(uid, rk, n)
                if args.comm_shared:
                    for q in rs[1:]:
                        q.comm_share(rs[0])
            ok, _ = agree_on_libpt_collective(dist, rank, world, pa.Renderer.comm_unique_id, init_rank,
                                              own_world_of_one=rehearsal)
            if ok and not args.comm_shared:           # further communicators, one per remaining context
                for q in rs[1:]:
                    ok = ok and agree_on_libpt_collective(dist, rank, world, pa.Renderer.comm_unique_id,
                                                          lambda uid, rk, n, q=q: q.comm_init_rank(uid, rk, n),
                                                          own_world_of_one=rehearsal)[0]
            if ok:
                comm_per_ctx = [q.comm_info() for q in rs]
                comm = comm_per_ctx[0]
            else:
                collective = "none (rehearsal on one GPU: frames are NOT combined)" if rehearsal else "torch"
        if collective == "torch":
            nccl_group = dist.new_group(backend="nccl")

    # PT_BENCH_COLLECTIVE_EXTRA_US (experiment, one-GPU boxes): a spin of that many microseconds queued behind the
    # all-reduce on the same stream -- a world of 1 reduces with a copy that takes no time once it runs, a world of 8
    # moves 2 x 7/8 x 24.9 MB over xGMI; the line printed says so and is not a measurement
    extra_us = float(os.environ.get("PT_BENCH_COLLECTIVE_EXTRA_US", "0"))

    def allreduce(q, cv):
        if collective == "libpt":
            q.allreduce_sum()                           # ncclAllReduce on the context's stream, into its total
            if extra_us > 0:
                torch.cuda._sleep(int(extra_us * 2400))
            return None
        if collective != "torch":
            return None                                 # (rehearsal without a collective)
        return dist.all_reduce(cv, op=dist.ReduceOp.SUM, group=nccl_group, async_op=True)

    pipe = PipelinedFrames(rs, canvases, rgbs, rank=rank, spp_total=args.spp, my_spp=my_spp,
                           mode=pa.PT_RGB8_MEAN_CLAMP, allreduce=allreduce if multi else None,
                           gate=args.gate) if pipelined else None

    def step():
        if pipe is not None:
            pipe.step()
            return
        r.reset()
        r.trace(my_spp)
        if multi:
            work = allreduce(r, canvas)
            if work is not None:
                work.wait()
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
        r = rs[(pipe.k - 1) % len(rs)]          # the context of the last frame: its counter holds one step
    segs_step = torch.tensor([r.segments], dtype=torch.float64)

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

    # N > 1: the collective on its own, event-timed on the launch stream, every rank entering together (a barrier
    # first, so the figure is the all-reduce and not the wait for the slowest rank's trace)
    coll_ms = []
    if multi and collective in ("libpt", "torch"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r_stream = pipe.mains[rs.index(r)] if pipe is not None else stream     # the stream context r launches on
        for _ in range(4):
            fence()
            with torch.cuda.stream(r_stream):
                e0.record(r_stream)
                work = allreduce(r, canvases[rs.index(r)])
                if work is not None:
                    work.wait()
                e1.record(r_stream)
            e1.synchronize()
            coll_ms.append(e0.elapsed_time(e1))
        coll_ms = coll_ms[1:]                           # the first one may set up RCCL's channels

    t_el = torch.tensor([elapsed], dtype=torch.float64)
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        mine = torch.tensor([elapsed / args.steps * 1e3, sum(coll_ms) / len(coll_ms) if coll_ms else 0.0], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x[0]) for x in every]
        rank_coll_ms = [float(x[1]) for x in every]
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
                       "scene_kernel_compiler": r.jit_info() if specialised else None,
                       "parallelism": f"sample-split x{world}",
                       "device_ordinal_rank0": local_rank,
                       "visible_devices": torch.cuda.device_count(),
                       "collective": None if not multi else
                                     (f"libpt ncclAllReduce behind the C ABI (pt_allreduce_sum), {comm.get('n_ranks')} RCCL ranks, "
                                      f"RCCL version code {comm.get('rccl_version')}, "
                                      + (f"one communicator per context ({len(rs)} per rank)" if not args.comm_shared else
                                         f"one communicator per rank shared by its {len(rs)} context(s) (pt_comm_share)")
                                      + ("; REHEARSAL on one GPU: a communicator of ONE rank per rank process, frames are NOT combined"
                                         if rehearsal else "") if collective == "libpt"
                                      else "torch.distributed all_reduce (nccl = RCCL)" if collective == "torch" else collective)
                                     + (f" + an artificial {extra_us:g}-us spin behind it (PT_BENCH_COLLECTIVE_EXTRA_US: NOT a measurement)"
                                        if multi and extra_us > 0 else ""),
                       "comm_info_per_context": comm_per_ctx if multi else None,
                       "per_rank_ms_per_step": None if dist is None else
                                               {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                                "all": [round(x, 4) for x in rank_ms]},
                       "collective_ms": None if not coll_ms else
                                        {"rank0_mean": round(sum(coll_ms) / len(coll_ms), 4), "rank0_min": round(min(coll_ms), 4),
                                         "per_rank_mean": [round(x, 4) for x in rank_coll_ms],
                                         "how": "the all-reduce alone after a barrier, HIP events on the launch stream, "
                                                "3 repetitions after one untimed; not part of the timed steps"},
                       "value_mode": "wall time of K pipelined frames (barrier + device synchronize on both sides, "
                                     "max over ranks); consecutive frames overlap on two streams"
                                     if pipe is not None else "wall time of K in-line frames",
                       "frame_pipeline": (f"{len(rs)} frames in flight ({len(rs)} canvases and trace streams): all-reduce + output "
                                          "of frame k overlap the trace of the next; "
                                          + ("frame k+1's trace starts when frame k's has dispatched its last workgroup (pt_trace_gate)"
                                             if pipe.gate else "no drain gate")) if pipe is not None else "none"},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                         "traffic": traffic,
                         "kernel": "trace_kernel" if (args.no_steal or args.no_primary_reuse) else
                                   ("trace_static_kernel" if len(scene) <= 8 else
                                    "trace_bvh_kernel" if indexed else "trace_steal_kernel"),
                         "kernel_ms": round(kern_ms, 4),
                         "kernel_ms_mode": "serial: a separate pass after the timed steps, one launch at a time, HIP "
                                           "events on the launch stream (pt_last_trace_ms), rank 0",
                         "traffic_source": None if traffic is None else
                                           "profiles/hbm_traffic.json (rocprofv3 PMC passes of an earlier run of this "
                                           "command, not measured in this run)",
                         "flops_per_segment": F, "segments_per_launch": int(segs_launch_rank0),
                         "note": "fp32 VALU-bound path (no MFMA, ~0 HBM bytes per segment); peak = fp32 vector "
                                 "157.3 TFLOP/s (= the fp32 MFMA peak)"},
        }
        if indexed:
            # trace_bvh_kernel's memory side is per-lane 16-byte gathers (one per index node visited, two or three per
            # object record), served by the CU's address unit and L1/L2 -- not HBM, not MFMA.  Algorithmic bytes per
            # launch = 16 B x the gathers the kernel counted (pt_bvh_counters); peak = the rate of fully divergent
            # 16-byte gathers measured on MI355X by tools/ubench_gather.hip (profiles/r03_bvh_walk.txt: 999 G
            # lane-gathers/s = 15,980 GB/s from an L1-resident table, 280 G distinct records/s = 4.5 TB/s from an
            # L2-resident one; the index's hot top levels sit in L1, ~90 % of its gathers hit there).  Since round 3
            # the kernel is bound by VALU issue under divergence (0.24 wave-instructions per cycle per SIMD of mostly
            # 4-cycle forms, half of the lanes active: profiles/r03_m_bvh_pmc_summary.txt), so frac says how far the
            # gather path is from ITS limit, not how far the kernel is from its own.
            nodes, recs = r.bvh_counters()
            pad = r.bvh_padding_gathers()               # the empty slot of odd leaves: fetched, not algorithmic
            fetched = (nodes + recs) / max(launches, 1)
            gathers = (nodes + recs - pad) / max(launches, 1)
            ach = gathers * 16 / (kern_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "l1-gather", "achieved": round(ach, 1), "peak": GATHER_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / GATHER_PEAK_GBS, 4), "traffic": None,
                               "kernel": "trace_bvh_kernel", "kernel_ms": round(kern_ms, 4),
                               "kernel_ms_mode": out["roofline"]["kernel_ms_mode"],
                               "gathers_per_segment": round(gathers / max(segs_launch_rank0, 1), 2),
                               "index_nodes_per_segment": round(nodes / max(launches, 1) / max(segs_launch_rank0, 1), 2),
                               "object_record_gathers_per_segment": round((recs - pad) / max(launches, 1) / max(segs_launch_rank0, 1), 2),
                               "padding_gathers_per_segment": round(pad / max(launches, 1) / max(segs_launch_rank0, 1), 2),
                               "fetched_gb_per_s": round(fetched * 16 / (kern_ms * 1e-3) / 1e9, 1),
                               "segments_per_launch": int(segs_launch_rank0),
                               "note": "memory side: divergent 16-byte per-lane gathers (index nodes and object records); the "
                                       "kernel itself is bound by VALU issue under divergence (DESIGN.md 3.4, round 3); "
                                       "algorithmic bytes = 16 B x gathers counted by the kernel "
                                       "(pt_bvh_counters) without the empty slot that leaves with an odd number of "
                                       "objects fetch (pt_bvh_padding_gathers; fetched_gb_per_s includes it); peak = measured rate of such gathers from an L1-resident "
                                       "16-KB table (tools/ubench_gather.hip, profiles/r03_bvh_walk.txt); the linear-scan "
                                       "flop formula of SURVEY 8d does not describe an indexed search"}
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
