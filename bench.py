"""Headline benchmark: Msamples/s of the render step on the 800x800 Cornell box.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: `spp` samples per pixel of
Scenes::cornell_2 at 800x800 (1024 spp, BASELINE.json configs[1]) rendered by
the HIP megakernel through the C ABI into a device-resident accumulation
buffer.  With N > 1 (one process per GPU: under torch.distributed.run, or, when
started plainly as `python bench.py --gpus N`, as N rank processes this script
launches itself before it touches the GPU)
every rank renders its own `spp` samples per step - disjoint global sample
numbers, so the ranks' images add up to one (N*spp)-sample image - and the
buffers are combined on rank 0 with one RCCL reduce per step (weak scaling).
The scene and camera are resident in HBM before the timed region.

Rank 0 prints one JSON line.  `roofline` prices the render kernel against the
FP32 vector peak (nothing here is a dense contraction, so there is no MFMA):
algorithmic flops = segments x (63 per triangle test + 18 per sphere test),
every surface counted as tested for every segment, divided by the kernel's
duration from HIP events recorded on the launch stream.  `cpu_baseline` times
the CPU oracle (a port of the reference's serial loop) on a bounded sample of
the same workload on this box's host, one thread (and, beside it as
`all_cores`, with OpenMP over image columns on up to 16 threads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

FP32_VECTOR_PEAK_TFLOPS = 157.3      # MI355X: 256 CU x 4 SIMD x 32 lanes x 2 flop (fma) x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell_2")
    ap.add_argument("--size", type=int, nargs=2, default=[800, 800], metavar=("W", "H"))
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per step per GPU")
    ap.add_argument("--numeric", choices=["exact", "fast"], default="exact")
    ap.add_argument("--layout", choices=["megakernel", "wavefront"], default="megakernel")
    ap.add_argument("--accel", choices=["auto", "none", "bvh", "tiles"], default="auto",
                    help="pt_set_accel mode: hierarchy by the library's rule, never, whenever possible, "
                         "or the linear scan with its records staged through LDS tiles")
    ap.add_argument("--partition", choices=["samples", "rows"], default="samples",
                    help="how N GPUs share a step: every rank its own samples of the whole image (the default), "
                         "or every rank all samples of its own band of rows (the step is then N x spp samples "
                         "per pixel on every rank's band, the same work in all)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=32,
                    help="samples per pixel of the CPU baseline's bounded sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(scene_name, width, height, spp):
    """The CPU oracle, one thread, on `spp` samples per pixel of the same scene."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pt_oracle
    sc = pt_oracle.builtin_scene(scene_name, width, height)
    pt_oracle.render(sc, 1, threads=1)                      # page in
    t0 = time.perf_counter()
    _, cnt = pt_oracle.render(sc, spp, first_sample=0, seed=1, threads=1)
    dt = time.perf_counter() - t0
    out = {"value": width * height * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1,
           "kind": "port",
           "sample": "%s %dx%d, %d spp (%.1f s, %.2f segments/sample)" % (
               scene_name, width, height, spp, dt, cnt.segments / cnt.samples)}
    # the same loop over columns with the OpenMP pragma the reference left commented out
    # (src/main.cpp:117), on this process's share of the host's cores
    share = len(os.sched_getaffinity(0))
    threads = max(1, min(share, int(os.environ.get("OMP_NUM_THREADS", "16")), 16))
    if threads > 1:
        many = 2 * spp
        t0 = time.perf_counter()
        pt_oracle.render(sc, many, first_sample=0, seed=1, threads=threads)
        dt = time.perf_counter() - t0
        out["all_cores"] = {"value": width * height * many / dt / 1e6, "unit": "Msamples/s", "cores": threads,
                            "sample": "%d spp (%.1f s)" % (many, dt)}
    return out


def intersection_kernel(pt_amd, scene_name, device):
    """The closest-hit search on its own: k_primary_hits over the camera rays of a 4096x4096
    image of the same scene (16.8 M rays, one segment each), priced like the render kernel."""
    side = 4096
    ctx = pt_amd.Context(device)
    ctx.load(pt_amd.builtin_scene(scene_name, side, side))
    out = {"kernel": "k_primary_hits", "rays": side * side}
    for name, numeric in (("exact", pt_amd.NUM_EXACT), ("fast", pt_amd.NUM_FAST)):
        best = None
        for _ in range(3):
            ctx.primary_hits(numeric)
            c = ctx.counters()
            if best is None or c.kernel_ms < best.kernel_ms:
                best = c
        tf = best.flops / (best.kernel_ms * 1e-3) / 1e12
        out[name] = {"kernel_ms": best.kernel_ms, "tests_per_s": best.tests / (best.kernel_ms * 1e-3),
                     "achieved": tf, "unit": "TFLOP/s", "frac": tf / FP32_VECTOR_PEAK_TFLOPS, "vgprs": int(best.vgprs)}
    ctx.close()
    return out


def kernel_source_hash():
    """SHA-1 over the device sources the render kernel is built from (sorted by name), the
    stamp tools/make_profile_summary.py stores beside the counters it collects."""
    import glob
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "path-tracer_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.hip"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_summary(kernel_ms):
    """The committed rocprofv3 PMC summary of the headline launch (tools/profile_gpu.sh +
    tools/make_profile_summary.py --traffic).  The counters come from a profiler run, not from
    this run, so they are reported only when they describe the kernel that just ran: the
    summary's source hash must equal the hash of today's device sources and its kernel time must
    lie within 5 % of this run's.  Returns (summary or {}, None or why it was not used)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return {}, "profiles/pmc_traffic.json is missing"
    if pmc.get("source_sha1") != kernel_source_hash():
        return {}, "profiles/pmc_traffic.json was collected on other device sources"
    ref_ms = pmc.get("kernel_ms")
    if not ref_ms or abs(kernel_ms - ref_ms) > 0.05 * ref_ms:
        return {}, "profiles/pmc_traffic.json: kernel %.2f ms then, %.2f ms now" % (ref_ms or 0.0, kernel_ms)
    return pmc, None


def launch_ranks(n):
    """`python bench.py --gpus N` started by hand: runs the same command line as N rank
    processes under torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) as a child of
    this process, which has not touched the GPU, and returns the child's exit code.  The
    ranks' output (rank 0's JSON line) passes through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    import pt_amd
    import pt_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("no GPU: the render step has no CPU fallback")
    # PT_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs
    # than ranks (ranks then share devices); the default is RCCL, one GPU per rank.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    # PT_BENCH_FORCE_PG=1: also at one rank, make the process group and run the per-step reduce
    # (torch's RCCL path end to end on a one-GPU box; rendezvous on 127.0.0.1 when nothing set it up)
    force_pg = os.environ.get("PT_BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        import datetime
        if "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        # a rank that dies before a collective must not leave the others waiting for the driver's
        # time limit: rendezvous and every collective give up after two minutes
        limit = datetime.timedelta(seconds=int(os.environ.get("PT_BENCH_PG_TIMEOUT", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=limit,
                                    device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)

    width, height = args.size
    mode = (pt_amd.MODE_WAVEFRONT if args.layout == "wavefront" else pt_amd.MODE_MEGAKERNEL) | \
           (pt_amd.NUM_FAST if args.numeric == "fast" else pt_amd.NUM_EXACT)
    scene = pt_amd.builtin_scene(args.scene, width, height)
    ctx = pt_amd.Context(device)
    ctx.set_accel({"auto": pt_amd.ACCEL_AUTO, "none": pt_amd.ACCEL_NONE, "bvh": pt_amd.ACCEL_BVH,
                   "tiles": pt_amd.ACCEL_LDS_TILES}[args.accel])
    ctx.load(scene)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    accum = torch.zeros(height, width, 3, dtype=torch.float32, device="cuda")

    def step(i):
        # step i renders global samples [i*world*spp, (i+1)*world*spp), split over the ranks
        if args.partition == "rows":
            pt_dist.render_band(ctx, accum, args.spp * world, world, rank,
                                first_sample=i * world * args.spp, seed=args.seed, mode=mode)
        else:
            pt_dist.render_share(ctx, accum, args.spp * world, world, rank,
                                 first_sample=i * world * args.spp, seed=args.seed, mode=mode)
        pt_dist.reduce_image(accum, dst=0, always=force_pg)    # one RCCL reduce; a no-op without a process group

    def fence():
        if world > 1 or force_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    kernel_ms, flops, segments = [], [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if world > 1 or args.partition == "rows":
            accum.zero_()                  # each step reduces one batch
        step(args.warmup + i)
        c = ctx.counters()                 # waits for this step's kernels (HIP events)
        kernel_ms.append(c.kernel_ms)
        flops.append(c.flops)
        segments.append(c.segments)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples_per_step = width * height * args.spp * world
        value = samples_per_step * args.steps / elapsed / 1e6
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = sum(flops) / len(flops) / (avg_ms * 1e-3) / 1e12
        n_surf = len(scene.surfaces)
        # counters belong to the headline launch only
        headline = (args.scene == "cornell_2" and (width, height, args.spp) == (800, 800, 1024) and
                    args.layout == "megakernel" and args.numeric == "exact")
        pmc, pmc_stale = pmc_summary(avg_ms) if headline else ({}, None)
        out = {
            "metric": "Msamples/sec, 800x800 Cornell box (diffuse+mirror+glass)",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp per GPU per step, %s, %s arithmetic" % (
                           args.scene, width, height, args.spp, args.layout, args.numeric),
                       "scene": args.scene, "width": width, "height": height, "spp_per_gpu": args.spp,
                       "layout": args.layout, "numeric": args.numeric, "accel": args.accel,
                       "parallelism": "%s x%d + one %s reduce per step" % (
                                          "row bands" if args.partition == "rows" else "spp-partition",
                                          world, "RCCL" if backend == "nccl" else backend)
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_VECTOR_PEAK_TFLOPS,
                         "traffic": pmc.get("hbm_bytes_per_launch"),
                         "kernel": ("k_wf_bounce (all launches)" if args.layout != "megakernel" else
                                    "k_render_mega_bvh" if c.accel == pt_amd.ACCEL_BVH else "k_render_mega"),
                         "kernel_ms": avg_ms,
                         "flops_per_launch": sum(flops) / len(flops)},
            "tests_per_s": sum(segments) / len(segments) * n_surf / (avg_ms * 1e-3),
            "segments_per_sample": sum(segments) / len(segments) / (width * height * args.spp),
            "redone_per_segment": c.redone / max(c.segments, 1),
            "vgprs": int(c.vgprs),
            "grid_blocks": int(c.grid_blocks),
        }
        if pmc_stale:
            out["roofline"]["traffic_stale"] = pmc_stale
        if force_pg:
            out["config"]["process_group"] = "%s, world %d" % (backend, world)
        if args.layout == "megakernel":
            # persistent workgroups of four waves, one per SIMD: workgroups per CU = waves per SIMD
            cus = torch.cuda.get_device_properties(device).multi_processor_count
            out["waves_per_simd"] = int(c.grid_blocks) // cus
        if "valu_insts_per_simd_cycle" in pmc:
            # the other reading of the same roofline: how busy the vector pipe is, whatever
            # the instructions compute (a SIMD issues at most one VALU instruction per two cycles)
            out["roofline"]["valu_issue"] = {"insts_per_simd_cycle": pmc["valu_insts_per_simd_cycle"],
                                             "peak": 0.5,
                                             "frac": pmc["valu_insts_per_simd_cycle"] / 0.5,
                                             "lane_utilisation": pmc.get("valu_lane_utilisation")}
        if c.accel == pt_amd.ACCEL_BVH:
            # flops and tests count what the reference's scan of every surface would
            # do for these segments, not what the traversal executed
            out["roofline"]["effective"] = True
        if world == 1 and args.layout == "megakernel":
            # the other numeric policy on the same workload, for the record (not the headline)
            other = "fast" if args.numeric == "exact" else "exact"
            omode = pt_amd.MODE_MEGAKERNEL | (pt_amd.NUM_FAST if other == "fast" else pt_amd.NUM_EXACT)
            ms = []
            for i in range(2):
                ctx.render_into(accum.data_ptr(), args.spp, (args.warmup + args.steps + i) * args.spp,
                                args.seed, omode)
                oc = ctx.counters()
                ms.append(oc.kernel_ms)
            out["other_numeric"] = {"numeric": other, "kernel_ms": min(ms),
                                    "value": width * height * args.spp / (min(ms) * 1e-3) / 1e6,
                                    "roofline_frac": oc.flops / (min(ms) * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS}
        if world == 1:
            out["intersection_kernel"] = intersection_kernel(pt_amd, args.scene, device)
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args.scene, width, height, args.cpu_spp)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # any rank's failure ends its process with a non-zero code at once (no interpreter
        # shutdown that could wait on a collective), so that torch.distributed.run tears the
        # other ranks down instead of leaving them to a time limit
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)
