#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + secondary) of the MI355X trace loop on the dragon scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one whole frame of the workload: bounding_boxes.yml (6 dragon instances of 23 490
triangles + translucent bounding cubes + reflective pedestals, 4 point lights) re-targeted to
1920x1080, reflection/refraction depth 5 (BASELINE.json configs[4], the config the metric is
quoted on; it fits one GPU).  The compiled scene and the framebuffer are resident in HBM before
the timed region starts.  With N > 1 the frame's 32x32 tiles are dealt round-robin to the ranks
(tile i -> rank i mod N), each rank traces its tiles into a compact buffer, and one RCCL gather
brings them to rank 0, which moves them into the [x][y][c] canvas.  Frames form a stream: frame k's gather
runs (asynchronously, on RCCL's stream) while frame k+1 is traced, and every frame is gathered and assembled
before the timed region ends.
Total work is fixed as N grows ("strong").  Rays are counted by the kernels, and only rays that are traced
count: a shadow query toward a light no cell of which contributes at the hit (it is behind the surface; the
reference's result is the ambient term whatever is_shadowed returns) is answered without a ray and reported
separately (`reference_equivalent_mrays_per_s` adds them back: the reference's loop traces them).

Timing: W untimed warm-up steps, then the loop of EXACTLY K timed steps -- barrier + device synchronisation on
both sides, wall clock, max over ranks -- is repeated until --min-seconds of timed work have been done (a single
K = 20 loop lasts 16 ms: too short for clocks to settle or for anybody outside to see the GPU busy).  The line
quotes the MEDIAN repetition and gives the spread.  Every launch sits between two HIP events on its own stream, so
`roofline.kernel_ms` is the average launch duration inside the very loop `ms_per_step` comes from.

The headline value is measured in fp64, the reference's arithmetic; the fp32 mode (half the registers and bytes) is
timed right after it and reported beside it (`fp32`), never as `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# per-work-item flop and byte prices (restated in DESIGN.md)
FLOPS_AABB, FLOPS_TRI, FLOPS_XFORM, FLOPS_ANALYTIC = 24, 57, 30, 40
FLOPS_SHADE, FLOPS_LIGHT, FLOPS_REFRACT = 85, 45, 40
PEAK_TFLOPS = {"fp64": 78.6, "fp32": 157.3}      # MI355X vector peaks (MI355X_MICROARCH.md; fp64 = half the fp32 rate)


def algorithmic_flops(work, wide=0):
    """`work` = rtc_stats::work; `wide` of its node visits were to four-wide triangle-BVH nodes (4 slab tests), the others to
    binary object-BVH nodes (2)."""
    node, tri, leaf, csg, shade, light, container, pattern = work
    return (FLOPS_AABB * (2 * (node - wide) + 4 * wide) + FLOPS_TRI * tri + (FLOPS_XFORM + FLOPS_ANALYTIC) * leaf + FLOPS_SHADE * shade +
            FLOPS_LIGHT * light + FLOPS_REFRACT * container)


def algorithmic_bytes(work, real_bytes, wide=0):
    node, tri, leaf = work[0] + wide, work[1], work[2]          # a four-wide record is two 64-byte halves
    # a node visit reads one node record -- priced at 64 bytes, the binary object-BVH record (the four-wide triangle-BVH record is
    # 112 of 128 bytes: the figure stays SURVEY.md 8d's so that rounds compare);
    # a triangle test p1,e1,e2,eps (10 reals); an analytic leaf test its 3x4 (12 reals)
    return node * 64 + tri * 10 * real_bytes + leaf * 12 * real_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="dragons")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the K-step timed loop until this much timed work is done (fp64; half of it for fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="upper bound on pixels traced by the CPU oracle")
    args = ap.parse_args()

    import torch
    from ray_tracer_challenge_amd import _ffi
    from ray_tracer_challenge_amd.compile import compile_scene
    from ray_tracer_challenge_amd.scene_io import load_world

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the trace loop)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    t_load = time.perf_counter()
    cam, scene_world, _ = load_world(os.path.join(ROOT, "scenes", args.scene + ".npz"))
    cam = cam.resized(args.width, args.height)
    t_compile = time.perf_counter()
    cs = compile_scene(scene_world)
    t_create = time.perf_counter()
    scene = _ffi.Scene(cs, device=local_rank)
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    info = scene.info()
    one_frame = {"load_scene_pack_ms": round((t_compile - t_load) * 1e3, 1), "compile_scene_ms": round((t_create - t_compile) * 1e3, 1),
                 "rtc_scene_create_ms": round((t_ready - t_create) * 1e3, 1)}

    from ray_tracer_challenge_amd.dist import TILE as T, FramePipeline, render_pipelined
    pipe = FramePipeline(cam.hsize, cam.vsize, world, rank, dev, T)      # N > 1: frame k's gather overlaps frame k+1's tracing
    fg = pipe.sets[0]
    rays_dev = torch.zeros(14, dtype=torch.int64, device=dev)
    frame = fg.frame

    def step(precision, count_work=False, feedback=True):
        stream = torch.cuda.current_stream().cuda_stream
        if world == 1:
            # whole frame straight into the canvas layout: no tile list, no gather
            scene.render_device(cam, frame.data_ptr(), rays_ptr=rays_dev.data_ptr(), precision=precision,
                                max_depth=args.depth, tile_size=T, stream=stream, count_work=count_work, feedback=feedback)
            return
        render_pipelined(scene, cam, pipe, precision, args.depth, rays_dev.data_ptr(), stream, count_work)

    def sync():
        if world > 1:
            pipe.drain()                # every frame submitted so far gathered and assembled on rank 0
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(precision, steps, warmup, feedback=True, min_seconds=0.0):
        """-> per-repetition wall seconds (max over ranks), rays of one repetition (summed over ranks), per-launch
        kernel ms of every repetition (this rank; HIP events on the launch stream around every step)."""
        for _ in range(warmup):
            step(precision, feedback=feedback)
        sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        reps, launches, rays_rep, total = [], [], None, 0.0
        while True:
            rays_dev.zero_()
            sync()
            t0 = time.perf_counter()
            for k in range(steps):
                ev[k][0].record()
                step(precision, feedback=feedback)
                ev[k][1].record()
            sync()
            dt = time.perf_counter() - t0
            dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
            rays_t = torch.cat([rays_dev[:4], rays_dev[12:13]])
            if dist is not None:
                dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
                dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
            reps.append(float(dt_t.item()))
            launches.append([a.elapsed_time(b) for a, b in ev])
            rays_rep = [int(v) for v in rays_t.tolist()]
            total += reps[-1]
            if total >= min_seconds or len(reps) >= 400:
                return reps, rays_rep, launches

    def median_rep(reps):
        order = sorted(range(len(reps)), key=lambda i: reps[i])
        return order[len(order) // 2]

    # the reference's own usage is ONE frame per scene (main.py:305-354): the first frame of a fresh handle, plain tile order
    if world == 1:
        t0 = time.perf_counter()
        step("fp64")
        torch.cuda.synchronize()
        one_frame["first_frame_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    reps64, rays, launches64 = timed("fp64", args.steps, args.warmup, min_seconds=args.min_seconds)
    launched64 = scene.last_launch()
    m64 = median_rep(reps64)
    dt = reps64[m64]
    total_rays = sum(rays[:4])            # rays actually traced; rays[4] = is_shadowed queries that needed none
    value = total_rays / dt / 1e6
    kms64 = float(np.mean(launches64[m64]))          # average launch duration inside the repetition `value` is quoted on
    # work counters from the instrumented variant (untimed), for the roofline numerator
    rays_dev.zero_()
    step("fp64", count_work=True)
    sync()
    work, wide = [int(v) for v in rays_dev[4:12].tolist()], int(rays_dev[13].item())
    reps32, rays32, launches32 = timed("fp32", args.steps, args.warmup, min_seconds=args.min_seconds / 2)
    launched32 = scene.last_launch()
    m32 = median_rep(reps32)
    dt32 = reps32[m32]
    kms32 = float(np.mean(launches32[m32]))
    rays_dev.zero_()
    step("fp32", count_work=True)
    sync()
    work32, wide32 = [int(v) for v in rays_dev[4:12].tolist()], int(rays_dev[13].item())
    # for the record: the same frames launched in plain tile order (no cost feedback from the previous frame)
    plain = None
    frame_host = ids_host = None
    if world == 1:
        n_plain = max(3, args.steps // 2)
        repsp, _, _ = timed("fp64", n_plain, 1, feedback=False)
        plain = repsp[0] / n_plain * 1e3
        # the frame the numbers are about, with hit ids, for the parity check against the CPU oracle's sample below
        ids_dev = torch.full((cam.hsize, cam.vsize), -9, dtype=torch.int32, device=dev)
        scene.render_device(cam, frame.data_ptr(), ids_ptr=ids_dev.data_ptr(), precision="fp64", max_depth=args.depth, tile_size=T,
                            stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        frame_host, ids_host = frame.cpu().numpy(), ids_dev.cpu().numpy()
        # the N = 1 frame through the multi-GPU path (tile list, compact buffer, tiles -> canvas): what a rank of N does, for one rank
        for _ in range(8):
            render_pipelined(scene, cam, pipe, "fp64", args.depth, rays_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
        sync_pipe = lambda: (pipe.drain(), torch.cuda.synchronize())
        sync_pipe()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            render_pipelined(scene, cam, pipe, "fp64", args.depth, rays_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
        sync_pipe()
        one_frame["pipelined_path_ms_per_frame_n1"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
        # and the whole drop-in call, as main.py makes it: render_multi(cam, world) -> ndarray (compile + upload + frame + copy back)
        from ray_tracer_challenge_amd.renderer import render_multi
        t0 = time.perf_counter()
        arr = render_multi(cam, scene_world, 1)
        one_frame["end_to_end_render_multi_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        one_frame["end_to_end_equals_timed_frame"] = bool(np.array_equal(arr, frame_host))

    # for the record: the same frame with the repeated mesh INSTANCED (compile_scene(instancing=True): one object-space dragon, six
    # instances) -- device memory and steady-state kernel ms, next to the flattened scene the value is quoted on
    instanced = None
    if world == 1:
        try:
            ics = compile_scene(scene_world, instancing=True)
            if ics.counts["n_instances"]:
                with _ffi.Scene(ics, device=local_rank) as isc:
                    rec = {"n_instances": int(ics.counts["n_instances"]), "mesh_triangles": int(ics.counts["n_mesh_tris"]),
                           "device_bytes_fp64": int(isc.info()["device_bytes_fp64"]), "device_bytes_fp64_flattened": int(info["device_bytes_fp64"])}
                    for prec in ("fp64", "fp32"):
                        ms = []
                        for _ in range(16):
                            st = {}
                            isc.render(cam, precision=prec, max_depth=args.depth, stats=st)      # (kernel_ms: the trace kernel alone, HIP events)
                            ms.append(st.get("kernel_ms"))
                        ms = [m for m in ms[8:] if m]
                        rec["kernel_ms_" + prec] = round(sorted(ms)[len(ms) // 2], 4) if ms else None
                    rec["policy"] = "compile_scene(instancing='auto') keeps this scene flattened: fewer than 250000 triangles would stay off the GPU, and the flattened form is the faster one in fp64"
                    instanced = rec
        except Exception as e:      # reporting only
            instanced = {"error": str(e)}

    # what every rank did: its share of the tiles, its kernel time per frame, and the gather on its own (N > 1)
    per_rank = None
    try:
        mine_ms = float(np.mean(launches64[m64]))
        gather_ms = 0.0
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync()
            e0.record()
            for _ in range(5):
                fg.gather()
            e1.record()
            torch.cuda.synchronize()
            gather_ms = e0.elapsed_time(e1) / 5
        row = torch.tensor([mine_ms, gather_ms, float(len(fg.mine))], dtype=torch.float64, device=dev)
        rows = [torch.zeros_like(row) for _ in range(world)]
        if dist is not None:
            dist.all_gather(rows, row)
        else:
            rows = [row]
        per_rank = [{"rank": r, "kernel_ms": round(float(v[0]), 4), "gather_ms": round(float(v[1]), 4), "tiles": int(v[2])} for r, v in enumerate(rows)]
    except Exception as e:          # noqa: BLE001 - reporting only
        per_rank = {"error": repr(e)}

    if rank == 0:
        flops = algorithmic_flops(work, wide)
        achieved = flops / (kms64 * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch_fp64")
        out = {
            "metric": "Mrays/sec (primary+secondary)", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "bounding_boxes.yml (6x dragon.obj, 140940 triangles + bbox cubes + pedestals, 4 lights) "
                                   "at %dx%d, depth %d" % (cam.hsize, cam.vsize, args.depth),
                       "scene_pack": args.scene, "tile": T, "parallelism": "tiles%d" % world,
                       "bvh_nodes": info["bvh_nodes"], "bvh_depth": info["bvh_depth"]},
            "repeats": {"of_the_timed_loop": len(reps64), "quoted": "median repetition", "timed_seconds": round(sum(reps64), 3),
                        "ms_per_step_min": round(min(reps64) / args.steps * 1e3, 4), "ms_per_step_max": round(max(reps64) / args.steps * 1e3, 4)},
            "rays_per_frame": {"primary": rays[0] // args.steps, "shadow": rays[1] // args.steps,
                               "reflection": rays[2] // args.steps, "refraction": rays[3] // args.steps,
                               "shadow_queries_answered_without_a_ray": rays[4] // args.steps},
            "reference_equivalent_mrays_per_s": round((total_rays + rays[4]) / dt / 1e6, 3),
            "ms_per_frame": round(dt / args.steps * 1e3, 4),
            "launch_order": {"policy": "dearest 8x8 blocks first, from the wave costs the first two frames of a view record; the dearest blocks split into 4x4 / 2x2 parts whose pixels are traced by 4 / 16 lanes sharing the shadow rays (DESIGN.md 4.1)",
                             "ms_per_frame_plain_tile_order": None if plain is None else round(plain, 4)},
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
            "per_rank": per_rank,
            "one_frame_per_scene": one_frame,
            "instanced": instanced,
            "fp32": {"value": round(sum(rays32[:4]) / dt32 / 1e6, 3), "unit": "Mrays/s", "ms_per_frame": round(dt32 / args.steps * 1e3, 4),
                     "kernel_ms": round(kms32, 4), "repeats": len(reps32), "kernel": launched32["kernel"],
                     "roofline": {"bound": "valu", "achieved": round(algorithmic_flops(work32, wide32) / (kms32 * 1e-3) / 1e12, 4),
                                  "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                                  "frac": round(algorithmic_flops(work32, wide32) / (kms32 * 1e-3) / 1e12 / PEAK_TFLOPS["fp32"], 5)}},
            "roofline": {"bound": "valu", "achieved": round(achieved, 4), "peak": PEAK_TFLOPS["fp64"], "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS["fp64"], 5), "traffic": traffic,
                         "traffic_source": None if traffic is None else "profiles/traffic.json (stored: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                                        "over this command, not measured in this run)",
                         "kernel": launched64["kernel"] + " (+ sum_rays_kernel, ~4 us, inside the same events)",
                         "launch": {k: launched64[k] for k in ("workgroup_threads", "workgroups", "n_units", "n_entries", "feedback_stage")},
                         "kernel_ms": round(kms64, 4),
                         "kernel_ms_source": "HIP events on the launch stream around each of the %d launches of the quoted repetition (mean); "
                                             "kernel_ms <= ms_per_step by construction" % args.steps,
                         "kernel_ms_median_of_all_launches": round(float(np.median(np.concatenate(launches64))), 4),
                         "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": algorithmic_bytes(work, 8, wide),
                         "algorithmic_bytes_note": "L1-level operand traffic of the counted work (node records, triangle rows, transforms); not HBM "
                                                   "bytes -- not comparable with `traffic`",
                         "work_per_launch": dict(zip(["bvh_node_visits", "triangle_tests", "analytic_leaf_tests", "csg_evals",
                                                      "shaded_hits", "light_samples", "container_passes", "pattern_evals"], work)),
                         "four_wide_node_visits": wide,
                         "flop_prices": "SURVEY.md 8d: slab test 24 (4 per four-wide node visit, 2 per binary one), Moller-Trumbore 57, ray -> object 30 + "
                                        "analytic leaf 40, shade 85, light sample 45, refraction / Schlick 40"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_on_sample"] = cpu_baseline(scene_world, cam, args.depth, args.cpu_sample, frame_host, ids_host)
        print(json.dumps(out), flush=True)
    scene.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(scene_world, cam, depth, n_sample, frame_host, ids_host):
    """The CPU oracle (a C port of the reference's trace loop: every leaf for every ray, as the
    reference does) on a bounded, evenly strided sample of the same frame's pixels -- timed as the CPU
    baseline and, since it has rendered those pixels anyway, compared with the GPU frame the bench timed."""
    from oracle.oracle import Oracle
    threads = min(16, os.cpu_count() or 1)
    n = cam.hsize * cam.vsize
    idx = (np.arange(n_sample, dtype=np.int64) * n) // n_sample + (n // n_sample) // 2
    idx = np.minimum(idx, n - 1)
    px, py = (idx % cam.hsize).astype(np.int32), (idx // cam.hsize).astype(np.int32)
    o = Oracle(scene_world)
    # chunks of the strided sample until ~12 s of CPU work have been done (bounded: <= n_sample pixels)
    order = np.random.default_rng(0).permutation(n_sample)
    px, py = px[order], py[order]
    rays = np.zeros(4, np.int64)
    done, dt, chunk = 0, 0.0, 256
    rgbs, leaves = [], []
    while done < n_sample and dt < 12.0:
        t0 = time.perf_counter()
        rgb, leaf, _, r = o.render_pixels(cam, px[done:done + chunk], py[done:done + chunk], remaining=depth, threads=threads)
        dt += time.perf_counter() - t0
        rays += r
        rgbs.append(rgb)
        leaves.append(leaf)
        done += min(chunk, n_sample - done)
        chunk = min(chunk * 2, 8192)
    n_sample = done
    base = {"value": round(float(rays.sum()) / dt / 1e6, 6), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%d evenly strided pixels of the same %dx%d frame (%d rays) in %.1f s; brute-force like the reference "
                      "(no bounding-box culling, shapes.py:454)" % (n_sample, cam.hsize, cam.vsize, int(rays.sum()), dt),
            "reference_python": reference_python_timing()}
    parity = None
    if frame_host is not None:
        want, leaf = np.concatenate(rgbs), np.concatenate(leaves)
        sx, sy = px[:n_sample], py[:n_sample]
        d = np.abs(frame_host[sx, sy] - want).max(axis=1)
        parity = {"pixels": int(n_sample), "max_abs": float(d.max()), "pixels_off_by_more_than_1e-9": int((d > 1e-9).sum()),
                  "id_mismatch": int((ids_host[sx, sy] != leaf).sum()),
                  "what": "the fp64 frame this bench timed against the CPU oracle's pixels of the cpu_baseline leg"}
    return base, parity


def reference_python_timing():
    """The reference's OWN Python trace loop, timed where it can run: in the build container, while it rendered the
    dragons fixture (oracle/make_golden.py records wall seconds and processes in the fixture's meta).  The rays of those
    pixels are counted here with the oracle, which traces the same rays (tests/test_oracle_golden.py)."""
    try:
        from oracle.oracle import Oracle
        from ray_tracer_challenge_amd.scene_io import load_world
        with np.load(os.path.join(ROOT, "tests", "golden", "dragons.npz"), allow_pickle=False) as z:
            meta = json.loads(z["meta"].tobytes().decode())
            gx, gy = z["px"], z["py"]
        cam, world, _ = load_world(os.path.join(ROOT, "scenes", "dragons.npz"))
        cam = cam.resized(meta["hsize"], meta["vsize"])
        _, _, _, rays = Oracle(world).render_pixels(cam, gx, gy, remaining=meta.get("remaining", 5))
        secs = float(meta["reference_seconds"])
        return {"value": round(float(rays.sum()) / secs / 1e6, 9), "unit": "Mrays/s", "cores": int(meta["procs"]), "kind": "reference",
                "sample": "%d pixels of the dragons scene at %dx%d (%d rays) in %.1f s on %d processes of the 8-core build container: "
                          "gbordelon/ray_tracer_challenge's own color_at, recorded when tests/golden/dragons.npz was made (the reference "
                          "cannot travel to the GPU box)" % (int(meta["pixels"]), meta["hsize"], meta["vsize"], int(rays.sum()), secs, int(meta["procs"]))}
    except Exception as e:          # noqa: BLE001 - a missing fixture must not take the bench line down
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
