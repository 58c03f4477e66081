#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s (primary + secondary) of Camera::render at 3840x2160.

Workload (BASELINE.json configs[1]): the reference's chapter-11a reflection/refraction scene
(rust/examples/chapter-11a.rs), recursion depth 5, 3840x2160, adaptive multisampling as the example ships it
(min 5 samples, allowed standard error 1e-2) on the seeded counter RNG.  A "step" is one full frame.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one of the ranks; started plainly
with --gpus N > 1 it starts torch.distributed.run itself as a child BEFORE anything touches the GPU and passes the
child's output and exit code on (the reference's fan-out needs no launcher either: camera.rs:260-264).
N > 1 shards the frame by 8-row tiles, round-robin over the ranks (strong scaling: the frame is fixed), and
gathers the rows to rank 0 with RCCL inside the timed region (raytracing_amd/tiles.py).  --frames-in-flight F (default: 2
on one GPU, 3 from 2 GPUs on, where a rank's share is short: tools/inflight.py, profiles/r03c_rank_share.txt) frames are
in flight at once, each on its own HIP stream with its own buffer (and its own work-queue slot inside the library): the
ends of one frame's kernels - a few waves finishing the deepest pixels - and its gather overlap the next frame's first
pass.  Every frame is rendered completely and gathered inside the timed region; F = 1 runs the frames strictly one after
the other.  Rank 0 prints ONE JSON line.  At N = 1 the line also carries `all_configs`: every other BASELINE
configuration measured live by this run after the headline's timed region (a few frames each; --no-all-configs skips it).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)": 256 CU x 128 lanes x 2 x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="frames rendered concurrently, one HIP stream each (the library has 4 work-queue slots per device); 0 = 2 on one GPU, 3 with 2 GPUs or more "
                         "(round 2's shares of the headline frame, ms per frame with 2 / 3 in flight: 1/2 of it 13.3 / 12.7, 1/4 7.56 / 6.68, 1/8 4.61 / 4.37; the whole frame 26.0 / 27.3)")
    ap.add_argument("--scene", default="ch11a", help="ch07 | ch11a | ch11b | teapot | soft | dof | dice (BASELINE configurations) | photons | yard | photonbox")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--sampling", default="shipped", choices=["shipped", "1spp"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--max-samples", type=int, default=0, help="cap on samples per pixel (0 = unbounded like the reference; diagnostics only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="headline only: skip the CPU baseline and, with it, the other extras that follow the timed region (all_configs, the host-frame measurement) - what the profiling scripts under tools/ pass")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N > 1 path with several ranks sharing one GPU (rows cross through host memory)")
    ap.add_argument("--gather", default="f32", choices=["f32", "rgb8"],
                    help="what crosses xGMI: the float canvas (default, what Camera::render returns) or Color::to_u8 bytes made on the device")
    ap.add_argument("--check-frame", action="store_true", help="rank 0 also renders the whole frame alone and compares it bit for bit with the gathered one")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded baseline sample")
    ap.add_argument("--no-all-configs", action="store_true", help="skip the live measurement of the other BASELINE configurations (N = 1 only)")
    ap.add_argument("--no-host-frame", action="store_true", help="skip the PCIe-inclusive measurement through rtmi_render (N = 1 only)")
    return ap.parse_args()


def self_launch(a):
    """--gpus N > 1 without a launcher: start the N ranks as a child `python -m torch.distributed.run ... bench.py <same
    arguments>` and pass its exit code on.  Nothing in THIS process has touched the GPU yet (torch is not even imported),
    and nothing is exec'ed: the child is an ordinary subprocess whose stdout (rank 0's JSON line) is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def workload_name(a):
    names = {"ch07": "chapter-07 spheres, hard shadows", "ch11a": "chapter-11a reflection/refraction, depth 5",
             "ch11b": "chapter-11b water reflection/refraction, depth 5", "teapot": "chapter-15 teapot mesh in host-built BVH",
             "soft": "soft_shadow sphere light", "dof": "depth-of-field thin lens", "dice": "chapter-16 CSG dice + lens",
             "photons": "photon_room with a map of 2 M photons, 100 nearest per estimate (not a BASELINE configuration)",
             "yard": "yard of every item kind (not a BASELINE configuration)",
             "photonbox": "examples/photon-map.rs room with a device-traced map of 2 M emitted photons, 100 nearest within 0.1 (not a BASELINE configuration)"}
    samp = "adaptive multisampling as shipped" if a.sampling == "shipped" else "1 centred sample per pixel"
    if a.max_samples:
        samp += " capped at %d samples/pixel" % a.max_samples
    return "%s, %dx%d, %s" % (names[a.scene], a.width, a.height, samp)


def host_threads():
    """The host cores this process may use: its CPU affinity, within the share the box gives a 1-GPU job (OMP_NUM_THREADS
    is set to that share there - 16 - while os.cpu_count() shows the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for var in ("RTMI_CPU_THREADS", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            return max(1, min(n, int(v)))
    return max(1, n)


def cpu_baseline(world, cam, a, scenes):
    """The oracle (a C/f64 port that keeps the reference's per-ray lists + recursion) timed on the host cores on a
    bounded sample of the same frame: every k-th row, k chosen so the run takes about --cpu-seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from raytracing_amd import abi
    threads = host_threads()
    opts = abi.make_opts(seed=a.seed, max_samples=a.max_samples)
    rays = lambda st: st["primary"] + st["reflection"] + st["refraction"] + st["shadow"]
    v = cam.vsize()
    stride = max(1, v // 16)  # probe: ~16 rows
    t0 = time.perf_counter()
    _, st = orc.render(world, cam, opts, n_threads=threads, row_stride=stride, row_offset=stride // 2)
    dt = time.perf_counter() - t0
    rows_probe = len(range(stride // 2, v, stride))
    per_row = dt / max(rows_probe, 1)
    rows_target = int(max(rows_probe, min(v, a.cpu_seconds / max(per_row, 1e-9))))
    if rows_target > rows_probe * 2:
        stride = max(1, v // rows_target)
        t0 = time.perf_counter()
        _, st = orc.render(world, cam, opts, n_threads=threads, row_stride=stride, row_offset=stride // 2)
        dt = time.perf_counter() - t0
    n_rows = len(range(stride // 2, v, stride))
    return {"value": rays(st) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "every %d-th row (%d of %d rows) of the same %dx%d frame, same seed; %.1f s wall on %d threads = all host cores "
                      "this job may use (the machine shows %d logical CPUs)" % (stride, n_rows, v, cam.hsize(), v, dt, threads, os.cpu_count() or 0)}


ALL_CONFIGS = (("ch07", "1spp", 8), ("ch07", "shipped", 4), ("ch11a", "1spp", 8), ("ch11b", "1spp", 8), ("ch11b", "shipped", 4), ("teapot", "1spp", 8),
               ("teapot", "shipped", 4), ("soft", "shipped", 4), ("dof", "shipped", 3), ("dice", "1spp", 6), ("dice", "shipped", 3))


def measure_config(torch, abi, scenes, dev, name, sampling, steps, a, F=2):
    """One more BASELINE configuration, measured like the headline (frames in flight on their own streams, wall time between
    synchronisations, every frame whole), plus the same call alone between HIP events for the dominant kernel's fraction."""
    world, cam = scenes.BASELINE_SCENES[name](a.width, a.height, deterministic=(sampling == "1spp"))
    scene = abi.Scene(world)
    opts = abi.make_opts(seed=a.seed)
    h, v = cam.hsize(), cam.vsize()
    bufs = [torch.zeros((v, h, 3), dtype=torch.float32, device=dev) for _ in range(F)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    stats = torch.zeros(abi.N_STATS, dtype=torch.int64, device=dev)

    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % F]):
                abi.render_device(scene, cam, opts, bufs[i % F], stats, streams[i % F].cuda_stream)
        torch.cuda.synchronize()
    run(max(F, 2))
    stats.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    S = {k: int(x) for k, x in zip(abi.STAT_NAMES, stats.cpu().tolist())}
    rays = S["primary"] + S["reflection"] + S["refraction"] + S["shadow"]
    one = torch.zeros(abi.N_STATS, dtype=torch.int64, device=dev)
    with torch.cuda.stream(streams[0]):
        m1, m2 = abi.render_device_timed(scene, cam, opts, bufs[0], one, streams[0].cuda_stream)
    O = {k: int(x) for k, x in zip(abi.STAT_NAMES, one.cpu().tolist())}
    fl2 = float(O["flops_second"]); fl1 = float(O["flops"]) - fl2
    dom_ms, dom_fl = (m1, fl1) if m1 >= m2 else (m2, fl2)
    scene.close()
    return {"ms_per_frame": round(dt / steps * 1e3, 3), "Mrays_s": round(rays / dt / 1e6), "rays": rays // steps, "samples": S["samples"] // steps,
            "steps": steps, "ms_call_alone": round(m1 + m2, 3), "ms_dominant_kernel": round(dom_ms, 3),
            "frac": round(dom_fl / (dom_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4) if dom_ms > 0 else 0.0,
            "frac_call_alone": round(float(O["flops"]) / ((m1 + m2) * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 4) if m1 + m2 > 0 else 0.0}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    from raytracing_amd import abi, scenes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world_size != a.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU path)")
    if a.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    xdev = dev if a.backend == "nccl" else torch.device("cpu")  # where the exchanged tensors live

    build = scenes.BASELINE_SCENES.get(a.scene) or scenes.EXTRA_SCENES[a.scene]
    world, cam = build(a.width, a.height, deterministic=(a.sampling == "1spp"))
    F = a.frames_in_flight if a.frames_in_flight > 0 else (3 if world_size >= 2 else 2)
    os.environ.setdefault("RTMI_WORK_SLOTS_PREALLOC", str(F))  # one work-queue slot per stream, made with the scene: no allocation on the enqueue path
    scene = abi.Scene(world)
    prec = 0 if a.precision == "f32" else 1
    opts = abi.make_opts(seed=a.seed, max_samples=a.max_samples, precision=prec, rank=rank, nranks=world_size, tile_rows=a.tile_rows)
    h, v = cam.hsize(), cam.vsize()
    from raytracing_amd import tiles
    gather = tiles.RowGather(v, h, tile_rows=a.tile_rows, device=xdev, dtype=torch.uint8 if a.gather == "rgb8" else torch.float32)
    assert gather.rows == abi.rank_rows(v, rank, world_size, a.tile_rows)
    NB = max(F, 2)  # buffers: at least two, so that frame i+1 renders while frame i is gathered even with one stream
    bufs = [torch.zeros((gather.padded_rows, h, 3), dtype=torch.float32, device=dev) for _ in range(NB)]
    bufs8 = [torch.zeros((gather.padded_rows, h, 3), dtype=torch.uint8, device=dev) for _ in range(NB)] if a.gather == "rgb8" else None
    stats = torch.zeros(abi.N_STATS, dtype=torch.int64, device=dev)
    render_streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    ev_rendered = [torch.cuda.Event() for _ in range(NB)]
    ev_free = [torch.cuda.Event() for _ in range(NB)]
    frames = {"n": 0, "last": None}

    def step(timed_events=None):
        """One frame: render this rank's row tiles (own stream), then gather the rows to rank 0 and assemble the canvas."""
        b = frames["n"] % NB
        render_stream = render_streams[frames["n"] % F]
        frames["n"] += 1
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(render_stream):
            render_stream.wait_event(ev_free[b])
            if timed_events is not None:
                timed_events[0].record(render_stream)
            abi.render_device(scene, cam, opts, bufs[b], stats, render_stream.cuda_stream)
            if timed_events is not None:
                timed_events[1].record(render_stream)
            ev_rendered[b].record(render_stream)
        cur.wait_event(ev_rendered[b])
        send = bufs[b]
        if a.gather == "rgb8":  # Color::to_u8 on the device (color.rs:53-60): 3 B per pixel over the links instead of 12
            abi.tonemap_u8_device(bufs[b], bufs8[b], cur.cuda_stream)
            send = bufs8[b]
        frames["last"] = gather.gather(send if a.backend == "nccl" else send.cpu())
        ev_free[b].record(cur)

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    stats.zero_()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(pairs[i])
    barrier()
    dt = time.perf_counter() - t0
    # per launch: the span of the timed region on the render streams / launches (with F > 1 the launches overlap, so a
    # single launch's own start-to-end time says little), and afterwards the same call alone on an idle GPU
    span_ms = max(pairs[0][0].elapsed_time(p[1]) for p in pairs)
    # ... one call alone, three times, with HIP events on its stream before, between and after the two passes
    # (rtmi_render_device_timed) and counters of its own: the per-kernel figures of the roofline object
    serial, first_ms, second_ms = [], [], []
    one = torch.zeros(abi.N_STATS, dtype=torch.int64, device=dev)
    for _ in range(3):
        one.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(render_streams[0]):
            m1, m2 = abi.render_device_timed(scene, cam, opts, bufs[0], one, render_streams[0].cuda_stream)
        serial.append(m1 + m2); first_ms.append(m1); second_ms.append(m2)
    serial_ms = sum(serial) / len(serial)
    one_call = {k: int(x) for k, x in zip(abi.STAT_NAMES, one.cpu().tolist())}

    t = torch.tensor([dt], dtype=torch.float64, device=xdev)
    tot = stats.clone().to(xdev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    tot = tot.cpu().numpy()
    st_local = stats.cpu().numpy()
    S = {k: int(tot[i]) for i, k in enumerate(abi.STAT_NAMES)}
    rays = S["primary"] + S["reflection"] + S["refraction"] + S["shadow"]

    if a.check_frame and rank == 0:
        whole = torch.zeros((v, h, 3), dtype=torch.float32, device=dev)
        abi.render_device(scene, cam, abi.make_opts(seed=a.seed, max_samples=a.max_samples, precision=prec), whole, None,
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if a.gather == "rgb8":
            whole8 = torch.zeros((v, h, 3), dtype=torch.uint8, device=dev)
            abi.tonemap_u8_device(whole, whole8, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            whole = whole8
        same = bool(torch.equal(whole.to(frames["last"].device), frames["last"]))
        print("check-frame: gathered frame %s the single-rank frame" % ("EQUALS" if same else "DIFFERS FROM"), file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)
    if rank == 0:
        k_ms = span_ms / a.steps
        flops_launch = float(st_local[abi.STAT_NAMES.index("flops")]) / a.steps  # this rank's launch
        achieved = flops_launch / (k_ms * 1e-3) / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                with open(tp) as f:
                    traffic = json.load(f).get("%s_%s" % (a.scene, a.sampling))
            except Exception:
                traffic = None
        ms1, ms2 = sum(first_ms) / 3.0, sum(second_ms) / 3.0
        fl2 = float(one_call["flops_second"])
        fl1 = float(one_call["flops"]) - fl2
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        tree = "tree" if a.scene in ("teapot", "dice", "yard") else ("tree, photon map" if a.scene == "photons" else "flat")
        kernels = {
            "render_kernel<%s, %s> (first pass)" % (a.precision, tree): {"ms": ms1, "algorithmic_flops": fl1, "achieved": tf(fl1, ms1)},
            "resample_kernel<%s, %s> + units_kernel + sum_up_kernel (second pass)" % (a.precision, tree): {"ms": ms2, "algorithmic_flops": fl2, "achieved": tf(fl2, ms2)},
        }
        for kv in kernels.values():
            kv["frac"] = kv["achieved"] / PEAK_FP32_VALU_TFLOPS
        dominant = max(kernels, key=lambda k: kernels[k]["ms"])
        # VALU busy %, lane utilisation and algorithmic FLOP per active lane-op of the dominant kernel: from the committed PMC
        # passes of this build (profiles/pmc_summary.json, written by tools/pmc_summary.py; counters cannot be read in-process)
        pmc = None
        tp = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(tp):
            try:
                with open(tp) as f:
                    pmc = json.load(f).get("%s_%s" % (a.scene, a.sampling))
            except Exception:
                pmc = None
        out = {
            "metric": "Mrays/s (primary+secondary) at 3840x2160, 1/2/4/8 MI355X; % FP32 VALU peak",
            "value": rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": a.precision,
            "data": "synthetic",
            "config": {"workload": workload_name(a), "scene": a.scene, "max_depth": int(world.max_reflection_depth),
                       "seed": a.seed, "sharding": ("8-row tiles round-robin, %s gather of %s rows to rank 0" % (a.backend, a.gather)) if world_size > 1 else "none",
                       "frames_in_flight": F,
                       "rays_per_step": rays // a.steps, "samples_per_step": S["samples"] // a.steps,
                       "discarded_samples_per_step": S["discarded"] // a.steps,
                       "ray_classes_per_step": {k: S[k] // a.steps for k in ("primary", "reflection", "refraction", "shadow")}},
            "roofline": {"bound": "valu", "kernel": dominant, "achieved": kernels[dominant]["achieved"], "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": kernels[dominant]["frac"], "traffic": traffic,
                         "valu_busy": pmc and pmc.get("valu_busy"), "lane_util": pmc and pmc.get("lane_util"),
                         "flop_per_lane_op": pmc and pmc.get("flop_per_lane_op"), "valu_insts_per_wave_ray": pmc and pmc.get("valu_insts_per_wave_ray"),
                         "pmc_source": pmc and pmc.get("source"),
                         "per_kernel": kernels,
                         "call": {"ms_alone": serial_ms, "frac_alone": flops_launch / (serial_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                                  "ms_in_flight": k_ms, "frac_in_flight": achieved / PEAK_FP32_VALU_TFLOPS,
                                  "algorithmic_flops": flops_launch},
                         "note": "FP32 vector (VALU) roofline, no MFMA on this path.  achieved / frac = the DOMINANT kernel of one "
                                 "rtmi_render_device call (rank 0's share): its algorithmic FLOPs (reference arithmetic per event, DESIGN.md "
                                 "'Roofline and accounting') / its duration between HIP events recorded on the launch stream before and after "
                                 "it (rtmi_render_device_timed, the call alone on the GPU, mean of 3, after the timed region); per_kernel has "
                                 "both passes; call.* is the whole call: alone, and as the span of the timed region / calls with "
                                 "frames_in_flight calls overlapping; traffic = HBM bytes per call from PMC passes (profiles/traffic.json); "
                                 "valu_busy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / GRBM_GUI_ACTIVE per XCD, lane_util = SQ_THREAD_CYCLES_VALU / (64 x "
                                 "SQ_ACTIVE_INST_VALU), flop_per_lane_op = algorithmic FLOPs / SQ_THREAD_CYCLES_VALU (active lanes summed over VALU instructions) of the dominant kernel, "
                                 "from the committed PMC passes named in pmc_source (profiles/pmc_summary.json)"},
        }
        if world_size == 1 and not a.no_host_frame and not a.no_cpu_baseline:
            # The drop-in call a foreign-language front end makes: rtmi_render, host Canvas out (camera.rs:155-161).  Kernels + the
            # device->host copy of the frame; never `value`.
            abi.render_host(scene, cam, abi.make_opts(seed=a.seed, max_samples=a.max_samples, precision=prec))
            t1 = time.perf_counter()
            n_host = 3
            for _ in range(n_host):
                _, hst = abi.render_host(scene, cam, abi.make_opts(seed=a.seed, max_samples=a.max_samples, precision=prec))
            ht = (time.perf_counter() - t1) / n_host
            hrays = sum(hst[k] for k in ("primary", "reflection", "refraction", "shadow"))
            out["pcie_inclusive"] = {"entry_point": "rtmi_render (host float buffer out)", "ms_per_frame": ht * 1e3, "Mrays_s": hrays / ht / 1e6,
                                     "frames": n_host, "note": "one frame at a time; kernels + device->host copy of the 12 B/pixel frame"}
        if world_size == 1 and not a.no_all_configs and not a.no_cpu_baseline and (a.width, a.height, a.precision, a.max_samples) == (3840, 2160, "f32", 0):
            rows = {}
            for name, sampling, n in ALL_CONFIGS:
                if (name, sampling) == (a.scene, a.sampling):
                    continue
                try:
                    rows["%s_%s" % (name, sampling)] = measure_config(torch, abi, scenes, dev, name, sampling, n, a)
                except Exception as e:  # (a row that fails is reported, not hidden)
                    rows["%s_%s" % (name, sampling)] = {"error": str(e)}
            out["all_configs"] = {"_what": "the other BASELINE configurations at 3840x2160, f32, measured LIVE by this run after the headline's timed "
                                           "region: 2 frames in flight, wall time over `steps` whole frames; frac = the dominant kernel of one call "
                                           "alone between HIP events", "rows": rows}
        if not a.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(world, cam, a, scenes)
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()


if __name__ == "__main__":
    main()
