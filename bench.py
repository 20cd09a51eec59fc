#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the Cornell-box render at 1920x1080, 64 spp on N MI355X.

One step = one complete render of the frame: every rank renders its tiles (tile t belongs to rank
t % N, rt_render_device on the rank's GPU), the tile buffers are gathered on rank 0 with RCCL
(torch.distributed, backend nccl) and scattered into the row-major image there; the gather of one frame
runs while the next frame renders (two tile buffers), and the last frame is flushed before the clock stops.  The scene lives
in HBM before the timed region starts; nothing crosses PCIe inside it.  `value` counts every
closest-hit query (primary, shadow, reflected, refracted) of all ranks per second of wall time
(max over ranks, barrier + synchronize on both sides).

The same JSON line carries
  roofline      the render kernel against the FP32 vector peak: algorithmic FLOP per launch
                (rays per launch x FLOP per ray, the latter counted by the CPU oracle with the
                constants of include/rt_flops.h) / the kernel's mean duration (HIP events on the
                launch stream)
  cpu_baseline  the CPU oracle (a port of the reference algorithm, one thread like the reference)
                timed on this host on a bounded sample of the same workload (N = 1 only)
Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 through torch.distributed.run.
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
# Memory-side traffic of one k_render launch of the default workload, bytes: rocprofv3 FETCH_SIZE + WRITE_SIZE of this
# build (profiles/r1_h_traffic.txt; counters cannot be read from inside the process).  Almost all of it is register-spill
# and path-frame traffic to scratch memory that stays in the Infinity Cache; the image itself is 25 MB.
MEASURED_TRAFFIC_BYTES_1080P_64SPP = 11.81e9


def cpu_baseline(w, h, spp, seed, threads=1):
    """Oracle render of the same frame at reduced spp on the host cores; returns the record for the
    JSON line and the algorithmic FLOP per ray of the workload."""
    from oracle import orc
    scene = orc.Scene.builtin(2)
    cam = orc.camera_default(w, h)
    scene.reset_counters()
    t0 = time.perf_counter()
    _, st = scene.render(cam, w, h, spp, orc.RT_RNG_COUNTER, seed, threads=threads)
    dt = time.perf_counter() - t0
    counters, flops = scene.counters()
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
    return rec, flops / st.rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=3, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--flop-per-ray", type=float, default=0.0, help="override instead of counting on the CPU")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracing_amd import api, host, parallel_render

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

    w, h, spp, tw, th = args.width, args.height, args.spp, args.tile, args.tile
    scene = host.HostScene.builtin(2).upload()          # Cornell box, built by the product's host code
    cam = host.camera_default(w, h)
    counters = parallel_render.new_counters(dev)
    renderer = parallel_render.ShardedRenderer(
        w, h, tw, th, dev, parallel_render.device_render_fn(scene, cam, w, h, spp, args.seed, tw, th, counters))
    stream = torch.cuda.current_stream(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    image = None

    def step(i=None):
        nonlocal image
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
    for i in range(args.steps):
        tick["i"] = i
        step(i)
    last = renderer.flush()  # every one of the K frames is complete (gathered and scattered) inside the timed region
    image = last if last is not None else image
    fence()
    dt = time.perf_counter() - t0

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    totals = counters.sum(0)
    stats = torch.tensor([float(totals[0].item()), float(totals[1].item()), dt, kernel_ms, float(totals[5].item())], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rays_total, samples_total, dt_max, kernel_ms_max, skipped_total = tsum[0].item(), tsum[1].item(), tmax[2].item(), tmax[3].item(), tsum[4].item()
    else:
        rays_total, samples_total, dt_max, kernel_ms_max, skipped_total = stats[0].item(), stats[1].item(), dt, kernel_ms, stats[4].item()
    # rays_total counts what the reference does: one per Scene::computeIntersection call.  The shadow queries of glass nodes,
    # whose Phong colour the reference computes and then overwrites, are counted there but not traced: `value` is on the
    # rays actually traced, the roofline numerator (algorithmic work) on the reference's count.
    traced_total = rays_total - skipped_total

    if rank == 0:
        img = image
        assert torch.isfinite(img).all() and float(img.mean()) > 0.01, "render produced an empty / non-finite image"
        cpu_rec, flop_per_ray = None, args.flop_per_ray
        if world == 1 and args.cpu_spp > 0:
            cpu_rec, counted = cpu_baseline(w, h, args.cpu_spp, args.seed)
            if flop_per_ray <= 0:
                flop_per_ray = counted
        if flop_per_ray <= 0:
            flop_per_ray = 488.97  # what the oracle counts on this frame (include/rt_flops.h constants)
        rays_per_launch = rays_total / args.steps / world  # this rank's kernel ~ 1/world of the frame
        achieved = rays_per_launch * flop_per_ray / (kernel_ms_max * 1e-3) / 1e12
        line = {
            "metric": "Mrays/s (whole node) Cornell 1080p 64spp",
            "value": round(traced_total / dt_max / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"cornell_box(scene 2) {w}x{h} {spp}spp bounces10 light_samples16 counter-rng seed {args.seed}",
                       "tile": f"{tw}x{th}", "parallelism": f"tiles t%{world} per GPU + RCCL gather to rank 0",
                       "rays_traced_per_step": int(traced_total / args.steps), "reference_rays_per_step": int(rays_total / args.steps),
                       "reference_mrays_per_s": round(rays_total / dt_max / 1e6, 2),
                       "rays_per_sample": round(rays_total / samples_total, 3)},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4),
                         "traffic": (MEASURED_TRAFFIC_BYTES_1080P_64SPP / world
                                     if (w, h, spp) == (1920, 1080, 64) else None),
                         "traffic_note": "bytes per launch, FETCH_SIZE + WRITE_SIZE from profiles/r1_h_traffic.txt (scratch, "
                                         "Infinity-Cache resident); algorithmic output 12 B per pixel",
                         "kernel": "k_render", "kernel_ms": round(kernel_ms_max, 3), "flop_per_ray": round(flop_per_ray, 2),
                         "kernel_mrays_per_s": round(rays_per_launch / (kernel_ms_max * 1e-3) / 1e6, 2),
                         "numerator": "reference-equivalent rays per launch x FLOP per ray (both counted by the CPU oracle's rules)"},
            "cpu_baseline": cpu_rec,
        }
        print(json.dumps(line), flush=True)
    fence()
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
