"""Headline benchmark: Mrays/s of the Cornell-box render at 1920x1080, 64 spp on N MI355X (--config c2, the default);
--config c1 | c3 | c4 | c5 runs the other workloads of raytracing_amd/workloads.py through the same loop.

One step = one complete render of the frame: every rank renders its tiles (tile t belongs to rank
t % N, rt_render_device on the rank's GPU; c5: four progressive passes of 16 spp with rt_render_pass_device),
the tile buffers are gathered on rank 0 with RCCL
(torch.distributed, backend nccl) and scattered into the row-major image there; the gather of one frame
runs while the next frame renders (two tile buffers), and the last frame is flushed before the clock stops.  The scene lives
in HBM before the timed region starts; nothing crosses PCIe inside it.  `value` counts every
closest-hit query traced (primary, shadow, reflected, refracted) of all ranks per second of wall time
(max over ranks, barrier + synchronize on both sides).

The same JSON line carries
  roofline      c1 / c2 (7 objects, the scene lives in the scalar cache): the render kernel against the FP32 vector
                peak.  `frac` counts the work of the rays the kernel traces: rays traced per launch x algorithmic FLOP per
                traced ray (counted by the CPU oracle with the constants of include/rt_flops.h, the Phong evaluations whose
                colour the reference overwrites left out) / the kernel's mean duration (HIP events on the launch stream);
                `frac_reference_work` counts everything the reference executes, discarded work included.  The reference
                arithmetic has no fused multiply-add, so 0.5 of the peak is the ceiling (`ceiling_no_fma`).
                c3 .. c5 (triangle BVH): what the hardware counters of the same kernel sources say (profiles/counters_cX.json,
                rocprofv3 --pmc, reported only while the kernel sources are unchanged): memory-side bytes per launch / the
                live kernel duration against the HBM peak, L2 request rate and hit rate, VALU issue share and active lanes;
                the logical bytes (64 B per BVH node fetched + 48 B per triangle tested + 32 B per ray + 12 B per sample)
                are given as `algorithmic_bytes`, not as a fraction of anything.
  cpu_baseline  the CPU oracle (a port of the reference algorithm, one thread like the reference)
                timed on this host on a bounded sample of the same workload (N = 1 only)
Launch: python bench.py [--gpus N --steps K --warmup W --config cX]; for N > 1 through torch.distributed.run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X spec: 256 CU x 4 SIMD x 16 lanes x 2 (packed) x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E
# Hardware counters (memory-side traffic, L2, SQ) cannot be read from inside the process: they are measured offline with
# tools/prof_round3.sh on the default frame of each workload and kept in profiles/counters_<config>.json with the hash of the
# kernel sources; the line reports them only while that hash still matches.


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("rt_device.h", "rt_kernels.h", "rt_types.h"):
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def profiled_counters(config, world):
    """(record, note): profiles/counters_<config>.json (tools/counters_json.py: rocprofv3 --pmc passes of one frame) if it
    was measured on the current kernel sources, else (None, why not)."""
    path = os.path.join(ROOT, "profiles", f"counters_{config}.json")
    if world != 1 or not os.path.exists(path):
        return None, "not measured for this configuration / rank count"
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{os.path.relpath(path, ROOT)} was measured on other kernel sources ({rec.get('kernel_source_hash')}); not reported"
    return rec, (f"profiled offline: one rocprofv3 --pmc run per counter set over one frame, {os.path.relpath(path, ROOT)}; FETCH_SIZE / WRITE_SIZE are "
                 "memory-side requests of the L2 (Infinity-Cache hits included); fetch_bytes_x2 applies the gfx950 rule that a 128-B read request is "
                 "tallied as 64 B, an upper bound for the dword-per-lane scratch reloads that dominate here")


def counter_fields(rec, kernel_s):
    """The counter-derived figures of one launch, rates taken with the live kernel duration."""
    if rec is None:
        return {}
    out = {"fetch_bytes": rec["fetch_bytes"], "write_bytes": rec["write_bytes"], "fetch_bytes_x2": 2 * rec["fetch_bytes"],
           "fabric_gbs": round((2 * rec["fetch_bytes"] + rec["write_bytes"]) / kernel_s / 1e9, 1)}
    if "TCC_REQ_sum" in rec:
        out["l2_request_gbs"] = round(rec["TCC_REQ_sum"] * 128 / kernel_s / 1e9, 1)
    for k in ("l2_hit_rate", "valu_issue_share", "active_lanes", "wait_inst_any_share", "wait_any_share", "valu_insts_per_64_traced_rays", "profiled_frame_ms"):
        if k in rec:
            out[k] = rec[k]
    return out


def cpu_baseline(w, h, spp, seed, threads=1):
    """Oracle render of the same frame at reduced spp on the host cores; returns the record for the
    JSON line and the algorithmic FLOP per ray of the workload: over everything the reference executes, and over the rays
    that remain when the Phong evaluations it discards (glass nodes) are left out."""
    from oracle import orc
    scene = orc.Scene.builtin(2)
    cam = orc.camera_default(w, h)
    scene.reset_counters()
    t0 = time.perf_counter()
    _, st = scene.render(cam, w, h, spp, orc.RT_RNG_COUNTER, seed, threads=threads)
    dt = time.perf_counter() - t0
    counters, flops = scene.counters()
    dead_rays, dead_flops = scene.discarded_work()
    rec = {"value": round(st.rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"same scene and camera, {w}x{h} at {spp} spp ({st.samples} samples, {st.rays} rays, {dt:.1f} s)",
           "rays_per_sample": round(st.rays / st.samples, 3)}
    # the reference is single-threaded; the counter RNG also allows row-parallel workers: the same sample on 16 cores
    many = min(16, os.cpu_count() or 1)
    if many > 1:
        t0 = time.perf_counter()
        _, st2 = scene.render(cam, w, h, spp, orc.RT_RNG_COUNTER, seed, threads=many)
        dt2 = time.perf_counter() - t0
        rec["all_cores"] = {"value": round(st2.rays / dt2 / 1e6, 3), "unit": "Mrays/s", "cores": many, "seconds": round(dt2, 2)}
    return rec, flops / st.rays, (flops - dead_flops) / (st.rays - dead_rays)


def cpu_baseline_mesh(wl, seed):
    """Mesh workloads: the oracle's Mesh::intersect is the reference's loop over all triangles, so the bounded sample is
    a set of random pixels of the same frame at 1 spp, sized for about 10 s on one core."""
    import numpy as np
    from oracle import orc
    from raytracing_amd import api
    scene = orc.Scene.from_desc(api.copy_struct(orc.SceneDesc, wl.desc), keep=wl)
    cam = orc.camera_default(wl.w, wl.h)
    n = max(4, int(1.5e9 / (25.0 * wl.desc.n_triangles)))   # ~25 rays per sample, ~7 ns per triangle test
    px = np.random.default_rng(seed).integers(0, wl.w * wl.h, n).astype(np.uint32)
    t0 = time.perf_counter()
    _, st = scene.render_pixels(cam, wl.w, wl.h, 1, px, seed=seed, threads=1)
    dt = time.perf_counter() - t0
    rec = {"value": round(st.rays / dt / 1e6, 6), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"same scene and camera, {n} random pixels of the {wl.w}x{wl.h} frame at 1 spp ({st.rays} rays, {dt:.1f} s; "
                     f"brute force over {wl.desc.n_triangles} triangles per ray, as the reference does)"}
    many = min(16, os.cpu_count() or 1)
    if many > 1:
        px2 = np.random.default_rng(seed + 1).integers(0, wl.w * wl.h, n * many // 2).astype(np.uint32)
        t0 = time.perf_counter()
        _, st2 = scene.render_pixels(cam, wl.w, wl.h, 1, px2, seed=seed, threads=many)
        dt2 = time.perf_counter() - t0
        rec["all_cores"] = {"value": round(st2.rays / dt2 / 1e6, 6), "unit": "Mrays/s", "cores": many, "seconds": round(dt2, 2)}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = 5 for c1..c3 and c5, 2 for c4")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--width", type=int, default=0, help="0 = the configuration's own")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--tile", type=int, default=8, help="tile edge in pixels (8: the shares of 8 ranks differ by 2 % on the Cornell frame, 16: by 4 %)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=3, help="spp of the bounded CPU-baseline sample of c1 / c2 (0 = skip the CPU baseline)")
    ap.add_argument("--flop-per-ray", type=float, default=0.0, help="override instead of counting on the CPU (over all reference rays)")
    ap.add_argument("--flop-per-traced-ray", type=float, default=0.0, help="the same over the rays traced (discarded Phong evaluations left out)")
    ap.add_argument("--native-gather", action="store_true", help="N > 1: exchange the tiles with the library's own RCCL calls (rt_gather_tiles_device)")
    ap.add_argument("--native-loop", action="store_true", help="every step is the library's own frame loop (rt_render_sharded_device: render, RCCL exchange and "
                                                               "un-tile enqueued by one C call, image left on rank 0's GPU) instead of the Python harness")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracing_amd import api, parallel_render, workloads

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    api.init(local_rank)
    if world > 1 and (args.native_gather or args.native_loop):
        box = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        api.comm_init(box[0], rank, world)

    # scene built by the product's host code; like the reference, Scene::setup_cornell_box_meshes announces the mesh it loaded
    # on stdout: sent to stderr here, stdout carries the JSON line only
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        wl = workloads.make(args.config)
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    w, h, spp, tw, th = args.width or wl.w, args.height or wl.h, args.spp or wl.spp, args.tile, args.tile
    pass_spp = min(wl.pass_spp, spp)
    steps = args.steps or (2 if args.config == "c4" else 5)
    mesh = wl.desc.n_triangles > 64
    t_build = time.perf_counter()
    scene = wl.upload()
    t_build = time.perf_counter() - t_build
    cam = wl.camera(w, h)
    counters = parallel_render.new_counters(dev)
    if pass_spp == spp:
        render_fn = parallel_render.device_render_fn(scene, cam, w, h, spp, args.seed, tw, th, counters)
    else:
        one_pass = parallel_render.device_pass_fn(scene, cam, w, h, args.seed, tw, th, counters)

        def render_fn(out, r, wd):  # progressive: the passes of one frame, sums staying in the tile buffer
            for s0 in range(0, spp, pass_spp):
                one_pass(out, r, wd, s0, min(pass_spp, spp - s0), spp)
    renderer = parallel_render.ShardedRenderer(w, h, tw, th, dev, render_fn, native_gather=args.native_gather and world > 1)
    stream = torch.cuda.current_stream(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    image = None

    d_image = torch.zeros(h * w * 3, dtype=torch.float32, device=dev) if (args.native_loop and rank == 0) else None
    if args.native_loop and pass_spp != spp:
        raise SystemExit("--native-loop renders a frame in one pass; this workload accumulates in passes")

    def step(i=None):
        nonlocal image
        if args.native_loop:  # one C call per frame; the HIP events bracket render + exchange + un-tile
            if i is not None:
                ev[i][0].record(stream)
            scene.render_sharded_device(cam, w, h, spp, args.seed, tw, th, d_image.data_ptr() if rank == 0 else 0, counters.data_ptr(), stream.cuda_stream)
            if i is not None:
                ev[i][1].record(stream)
            image = d_image.view(h, w, 3) if rank == 0 else None
            return
        out = renderer.step()  # pipelined for N > 1: returns the previous frame, flush() the last one
        image = out if out is not None else image

    # kernel-only timing: HIP events on the launch stream around the render call of each timed step
    inner = renderer.render_tiles_fn
    tick = {"i": None}

    def timed_render(out, r, wd):
        i = tick["i"]
        if i is not None:
            ev[i][0].record(stream)
        inner(out, r, wd)
        if i is not None:
            ev[i][1].record(stream)
    renderer.render_tiles_fn = timed_render

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    renderer.flush()
    fence()
    counters.zero_()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        tick["i"] = i
        step(i)
    last = renderer.flush()  # every one of the K frames is complete (gathered and scattered) inside the timed region
    image = last if last is not None else image
    fence()
    dt = time.perf_counter() - t0

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    totals = [float(v) for v in counters.sum(0).tolist()]   # rays, samples, nodes, bvh_nodes, bvh_tris, rays_skipped
    stats = torch.tensor(totals[:6] + [dt, kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rays_total, samples_total, _, bvh_nodes, bvh_tris, skipped_total = (tsum[k].item() for k in range(6))
        dt_max, kernel_ms_max = tmax[6].item(), tmax[7].item()
    else:
        rays_total, samples_total, _, bvh_nodes, bvh_tris, skipped_total = totals[:6]
        dt_max, kernel_ms_max = dt, kernel_ms
    # rays_total counts what the reference does: one per Scene::computeIntersection call.  The shadow queries of glass nodes,
    # whose Phong colour the reference computes and then overwrites, are counted there but not traced: `value` is on the
    # rays actually traced, and so is `roofline.frac`; `roofline.frac_reference_work` uses the reference's count.
    traced_total = rays_total - skipped_total

    if rank == 0:
        img = image
        assert torch.isfinite(img).all() and float(img.mean()) > 0.01, "render produced an empty / non-finite image"
        cpu_rec, flop_per_ray, flop_per_traced = None, args.flop_per_ray, args.flop_per_traced_ray
        if world == 1 and args.cpu_spp > 0:
            if mesh:
                cpu_rec = cpu_baseline_mesh(wl, args.seed)
            else:
                cpu_rec, counted, counted_traced = cpu_baseline(w, h, args.cpu_spp, args.seed)
                if flop_per_ray <= 0:
                    flop_per_ray = counted
                if flop_per_traced <= 0:
                    flop_per_traced = counted_traced
        if flop_per_ray <= 0:
            flop_per_ray = 488.97      # what the oracle counts on the c2 frame (include/rt_flops.h constants): per reference ray,
        if flop_per_traced <= 0:
            flop_per_traced = 479.43   # and per ray that is left when the discarded Phong evaluations of glass nodes are taken out
        counters_rec, counters_note = profiled_counters(args.config, world) if (w, h, spp) == (wl.w, wl.h, wl.spp) else (None, "non-default frame")
        per_launch = 1.0 / steps / world            # this rank's launches ~ 1 / world of the frame
        kernel_s = kernel_ms_max * 1e-3
        cf = counter_fields(counters_rec, kernel_s)
        traffic = (counters_rec["fetch_bytes"] + counters_rec["write_bytes"]) if counters_rec else None
        common = {"traffic": traffic, "traffic_note": counters_note, "counters": cf or None, "kernel": "k_render", "kernel_ms": round(kernel_ms_max, 3),
                  "kernel_mrays_per_s": round(traced_total * per_launch / kernel_s / 1e6, 2)}
        if mesh:
            # What bounds these kernels is read from the counters, not assumed: the walks are chains of dependent node fetches
            # (most of them L2 hits) in divergent code, far from the HBM rate.  Without counters of this build only the logical
            # bytes are known, and they are not a measure of memory traffic (32 B per ray + 12 B per sample never leave the
            # registers of a fused kernel), so no fraction is derived from them.
            alg_bytes = (bvh_nodes * 64 + bvh_tris * 48 + traced_total * 32 + samples_total * 12) * per_launch
            fabric = cf.get("fabric_gbs")
            roofline = {"bound": "latency+divergent_valu" if cf else "unclassified (no hardware counters of this build)",
                        "achieved": fabric, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fabric / HBM_PEAK_GBS, 4) if fabric else None,
                        "achieved_is": "memory-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE of the profiled frame) / live kernel duration",
                        "algorithmic_bytes": int(alg_bytes), "algorithmic_gbs": round(alg_bytes / kernel_s / 1e9, 1),
                        "algorithmic_bytes_are": "64 B per BVH node fetched + 48 B per triangle tested (kernel counters) + 32 B per ray + 12 B per sample",
                        "bytes_per_ray": round(alg_bytes / (traced_total * per_launch), 1),
                        "bvh_nodes_per_ray": round(bvh_nodes / traced_total, 3), "bvh_tris_per_ray": round(bvh_tris / traced_total, 3)}
        else:
            achieved = traced_total * per_launch * flop_per_traced / kernel_s / 1e12
            achieved_ref = rays_total * per_launch * flop_per_ray / kernel_s / 1e12
            roofline = {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4), "ceiling_no_fma": 0.5,
                        "numerator": "rays traced per launch x algorithmic FLOP per traced ray (CPU oracle's count, discarded Phong evaluations excluded; "
                                     "an upper bound of the arithmetic executed: any-hit shadow tests and culled squares do less than the reference's loops)",
                        "flop_per_traced_ray": round(flop_per_traced, 2),
                        "achieved_reference_work": round(achieved_ref, 3), "frac_reference_work": round(achieved_ref / FP32_VECTOR_PEAK_TFLOPS, 4),
                        "flop_per_reference_ray": round(flop_per_ray, 2)}
        roofline.update(common)
        names = {"c1": "Cornell 480x480 50spp", "c2": "Cornell 1080p 64spp", "c3": "100k-triangle mesh scene 1080p 64spp",
                 "c4": "1M-triangle mesh scene 4K 256spp", "c5": "10M-triangle scene 1080p 64spp progressive"}
        line = {
            "metric": f"Mrays/s (whole node) {names[args.config]}",
            "value": round(traced_total / dt_max / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {wl.what} {w}x{h} {spp}spp" + (f" in passes of {pass_spp}" if pass_spp != spp else "") +
                                   f" bounces10 light_samples16 counter-rng seed {args.seed}",
                       "tile": f"{tw}x{th}", "parallelism": f"tiles t%{world} per GPU + RCCL gather to rank 0" + (" (rt_gather_tiles_device)" if args.native_gather and world > 1 else "") + (" (library frame loop: rt_render_sharded_device)" if args.native_loop else ""),
                       "triangles": int(wl.desc.n_triangles), "scene_upload_s": round(t_build, 3),
                       "rays_traced_per_step": int(traced_total / steps), "reference_rays_per_step": int(rays_total / steps),
                       "reference_mrays_per_s": round(rays_total / dt_max / 1e6, 2),
                       "rays_per_sample": round(rays_total / samples_total, 3)},
            "roofline": roofline,
            "cpu_baseline": cpu_rec,
        }
        print(json.dumps(line), flush=True)
    fence()
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
