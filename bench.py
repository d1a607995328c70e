#!/usr/bin/env python3
"""bench.py - Mrays/s of the gfx950 rendering backend on BASELINE.json's workload.

A "step" is one pass of the hot path over one batch of synthetic input: one full
frame (demo scene 1 @1920x1080, reflections on, frozen at time 0) per GPU in the
job.  Scene data and the frame buffers are resident in HBM when the timed region
starts.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (quadray-engine_amd/sharding.py, DESIGN.md "Multi-GPU"): every step renders N
frames; each frame's 8-row tile rows are dealt to the N ranks in contiguous blocks
(rotated per frame for balance), every rank renders its blocks of all N frames, then
ONE exchange over RCCL moves the blocks so that rank f ends the step holding the
complete frame f.  Per-GPU work is constant in N (weak scaling); the exchange runs on
its own stream and overlaps the next step's kernels.  Three steps are in flight (--inflight),
each launch on its own HIP stream: a frame ends with a few long recursion waves, and the
next frame's bulk fills the GPU meanwhile (frames are independent, one target set per step in flight).

Rays are counted by the backend's counting kernel variant (not timed): primary +
shadow + reflection + refraction rays actually traced ("useful" rays: the backend
shades only the final hit of a list walk, the reference also shades overdrawn hits).
The same count is used for the CPU baseline so that the two Mrays/s are comparable.

Prints ONE JSON line on rank 0.
"""
import argparse
import gzip
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name -> (golden snapshot, reference-driver args for the CPU baseline, description)
    "demo1_1080p": ("c2b_demo01_1080p", ["--scene", "demo01", "-w", "1920", "-h", "1080"],
                    "demo scene 1 @1920x1080, reflections+refractions on (depth 10), t=0"),
    "demo1_1080p_d0": ("c2_demo01_1080p_d0", ["--scene", "demo01", "-w", "1920", "-h", "1080", "--depth", "0"],
                       "demo scene 1 @1920x1080, primary + hard-shadow rays only (depth 0)"),
    "demo2_1080p_gf_d3": ("c3_demo02_1080p_gf_d3",
                          ["--scene", "demo02", "-w", "1920", "-h", "1080", "--gamma", "--fresnel", "--depth", "3"],
                          "demo scene 2 @1920x1080, Gamma+Fresnel, depth 3"),
    "demo2_2160p_aa4": ("c4_demo02_2160p_aa4_gf",
                        ["--scene", "demo02", "-w", "3840", "-h", "2160", "--gamma", "--fresnel", "--fsaa", "4"],
                        "demo scene 2 @3840x2160, 4x FSAA, Gamma+Fresnel"),
    # a crowd built inside the engine (oracle/ref_driver.cpp --swarm): demo scene 1 + 240 quadrics in bounding-volume
    # arrays, the engine's own lists; the densest scene that is gated by a reference frame and timed against the reference
    "swarm_1080p": ("swarm_demo01_240_1080p", ["--scene", "demo01", "-w", "1920", "-h", "1080", "--swarm", "240,1"],
                    "demo scene 1 + a swarm of 240 spheres / bowls / ellipsoids (plain, metal, glass) @1920x1080, depth 10"),
    # BASELINE.json config 5: not a reference scene (quadray-engine_amd/synth.py builds the snapshot, the GPU
    # binning pass its tile lists); the CPU baseline is the oracle port on a sample of rows
    "synth10k_4320p": ("synth:10000:7680:4320:4", None,
                       "synthetic 10 000 quadrics @7680x4320, depth 4, 4 lights (quadray-engine_amd/synth.py)"),
    "synth10k_1080p": ("synth:10000:1920:1080:4", None,
                       "synthetic 10 000 quadrics @1920x1080, depth 4, 4 lights (quadray-engine_amd/synth.py)"),
}

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 (vector), counts an FMA as 2


def all_work(snap):
    """fp32 operations per frame (SURVEY.md 8(d) weights) of a workload, tests/golden/work.json: "deferred" = counted by the
    CPU oracle when the fixture was made (tests/golden/make_work.py), "kernel" = counted by the kernel's QR_PROF build
    (tools/gpu_work.py), "oracle_band" = the oracle's count on a band of rows where it cannot walk the whole frame."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "work.json")) as f:
            return json.load(f).get(snap, {})
    except Exception:
        return {}


def golden_hash(snap):
    """Reference frame fingerprint of a golden snapshot (tests/golden/manifest.json), or None."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
            return int(json.load(f)[snap]["hash"], 16)
    except Exception:
        return None


def committed_counters(workload):
    """Hardware-counter figures of the dominant kernel that a bench run cannot take itself (rocprofv3 --pmc
    passes, tools/gpu_profile_round.sh): per-launch SQ instruction counts and HBM bytes.  They come from
    profiles/counters.json and are tagged as NOT measured in this run, with the profile they came from."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            c = json.load(f).get(workload)
        if c is not None:
            c = dict(c)
            c["measured_in_this_run"] = False
            # quoted only beside a run of the build they were taken from (qr_version() carries the source fingerprint)
            from qr_loader import load_package
            this_lib = load_package().lib().qr_version().decode()
            if c.get("lib") != this_lib:
                c = dict(stale=True, lib=c.get("lib"), this_lib=this_lib, source=c.get("source"), measured_in_this_run=False)
        return c
    except Exception:
        return None


_SYNTH_CACHE = {}


def load_blob(name):
    if name.startswith("synth:"):
        if name not in _SYNTH_CACHE:
            n, w, h, d = (int(x) for x in name.split(":")[1:])
            spec = importlib.util.spec_from_file_location("qr_synth", os.path.join(ROOT, "quadray-engine_amd", "synth.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            # the generator writes the global hierarchical list only; per-object shadow lists come from the product's
            # list-building pass (qr_snapshot_build_lists_c: the role of the engine's ssort / lsort)
            from qr_loader import load_package
            _SYNTH_CACHE[name] = load_package().build_lists(mod.make_scene(n_objects=n, width=w, height=h, depth=d, shadow_lists=False))
        return _SYNTH_CACHE[name]
    with open(os.path.join(ROOT, "tests", "golden", name + ".qrs.gz"), "rb") as f:
        return gzip.decompress(f.read())


def load_sharding():
    spec = importlib.util.spec_from_file_location("qr_sharding", os.path.join(ROOT, "quadray-engine_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(workload, rays_per_frame, gpu_frame=None, frames_budget_s=12.0):
    """Time the CPU path on this box's host cores (rank 0, N=1 only): the unmodified reference
    (oracle/_ref/qr_ref, kind "reference") when its prebuilt binary travelled with the repo,
    else our OpenMP restatement (kind "port").  16 threads (where the reference's scanline interleave is
    fastest), with an all-cores and a single-thread figure beside it; `host_cores` is the box's core count."""
    total = os.cpu_count() or 1
    try:
        total = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(total, 16)     # the reference's row interleave is fastest around 16 threads; an all-cores figure goes beside it
    ref = os.path.join(ROOT, "oracle", "_ref", "qr_ref")
    snap, ref_args, _ = WORKLOADS[workload]
    if ref_args is None:
        # no reference scene: the oracle port on a horizontal band of the frame, rays and fp32 operations counted
        # by the oracle; the band's pixels are also the correctness check of the GPU frame for this workload
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import qr_oracle
        blob = load_blob(snap)
        h = qr_oracle.info(blob)["h"]
        rows = 8
        t0 = time.time(); _, _, c = qr_oracle.render(blob, threads=cores, rows=(h // 2, h // 2 + rows), deferred=True); dt = time.time() - t0
        rows = int(max(8, min(h // 2, rows * frames_budget_s / max(dt, 1e-3)))) // 8 * 8
        r0 = h // 2 - rows // 2
        t0 = time.time(); fr, _, c = qr_oracle.render(blob, threads=cores, rows=(r0, r0 + rows), deferred=True); dt = time.time() - t0
        rays = c["primary"] + c["shadow"] + c["reflect"] + c["refract"]
        res = dict(value=rays / dt / 1e6, unit="Mrays/s", cores=cores, host_cores=total, kind="port",
                   sample=f"{rows} rows around the middle of the frame with oracle/qr_oracle.c (scalar C + OpenMP, {cores} threads): "
                          f"{rays} rays in {dt:.2f} s",
                   band=dict(rows=[r0, r0 + rows], rays=rays, flops=c["flops"]))
        if gpu_frame is not None:
            res["band"]["gpu_pixels_differ"] = int((gpu_frame[r0:r0 + rows] != fr[r0:r0 + rows]).sum())
        return res
    if os.path.exists(ref):
        try:
            import tempfile
            tmp = tempfile.mkdtemp(prefix="qrbench_")
            os.makedirs(os.path.join(tmp, "dump"), exist_ok=True)

            def run(n, threads):
                out = subprocess.run([ref] + ref_args + ["--threads", str(threads), "--bench", str(n)],
                                     cwd=tmp, capture_output=True, text=True, timeout=300)
                line = [l for l in out.stdout.splitlines() if l.startswith("bench ")][0].split()
                head = out.stdout.splitlines()[0]
                return float(line[line.index("median_ms") + 1]), head

            def timed(threads, budget_s, cap):
                ms, _ = run(3, threads)             # calibrate, then fill the time budget
                n = int(max(5, min(cap, budget_s * 1000.0 / max(ms, 0.01))))
                ms, head = run(n, threads)
                return ms, n, head
            # three runs of a third of the budget each: the 16-thread figure of one run moved by +-20 % between boxes and between
            # runs on one box in round 3 (other tenants on the host); the middle run is reported, all three are listed
            runs = sorted(timed(cores, frames_budget_s / 3.0, 1000) for _ in range(3))
            ms, n, head = runs[1]
            simd = head.split("simd ")[1].split()[0] if "simd " in head else "auto"
            res = dict(value=rays_per_frame / ms / 1e3, unit="Mrays/s", cores=cores, host_cores=total, kind="reference",
                       sample=f"{n} frames of the same workload through the unmodified reference's rt_Scene::render "
                              f"(update phases included, SIMD target {simd}, {cores} threads on a box with {total} host cores), "
                              f"median {ms:.3f} ms/frame; the middle of three such runs",
                       runs_median_ms=[round(r[0], 4) for r in runs])
            for label, th, budget in (("all_cores", total, 6.0), ("single_thread", 1, 5.0)):
                if th == cores:
                    continue
                try:
                    ms1, n1, _ = timed(th, budget, 1000)
                    res[label] = dict(value=rays_per_frame / ms1 / 1e3, unit="Mrays/s", cores=th,
                                      sample=f"{n1} frames, median {ms1:.3f} ms/frame")
                except Exception as e:
                    print(f"{label} baseline failed: {e}", file=sys.stderr)
            return res
        except Exception as e:  # fall through to the port
            print(f"reference baseline failed: {e}", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qr_oracle
    blob = load_blob(snap)
    t0 = time.time(); qr_oracle.render(blob, threads=cores); dt = time.time() - t0
    n = int(max(2, min(200, frames_budget_s / max(dt, 1e-3))))
    ts = []
    for _ in range(n):
        t0 = time.time(); qr_oracle.render(blob, threads=cores); ts.append(time.time() - t0)
    ts.sort()
    ms = ts[len(ts) // 2] * 1e3
    return dict(value=rays_per_frame / ms / 1e3, unit="Mrays/s", cores=cores, host_cores=total, kind="port",
                sample=f"{n} frames of the same workload with oracle/qr_oracle.c (scalar C + OpenMP), median {ms:.3f} ms/frame")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="demo1_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3, help="steps (frames per GPU) in flight, each on its own streams")
    ap.add_argument("--min-region-ms", type=float, default=250.0,
                    help="repeat the --steps pass back to back until the timed region lasts at least this long")
    ap.add_argument("--repetitions", type=int, default=0, help="passes of --steps inside the timed region (0: from --min-region-ms)")
    ap.add_argument("--split-frame", action="store_true",
                    help="N > 1: ONE frame per step, its 8-row tile rows dealt round-robin over the N ranks and gathered on rank 0 "
                         "(strong scaling: north_star's 'tiles sharded across the GPUs, final image gathered'); default: N frames per step (weak)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: gather every frame on rank 0 (north_star's wording) instead of frame f on rank f")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qr_loader import load_package
    qr = load_package()
    sharding = load_sharding()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("QR_BENCH_SAME_DEVICE"):          # rehearsal of the N > 1 code path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("QR_BENCH_SAME_DEVICE"):
            dist.init_process_group("gloo")              # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    snap, _, desc = WORKLOADS[args.workload]
    blob = load_blob(snap)
    scn = qr.Scene(blob, device=local_rank, rebin_tiles=snap.startswith("synth:"))
    W, H = scn.width, scn.height
    n_groups = (H + 7) // 8
    N = world

    # deterministic ray count of one frame (counting kernel variant, not timed)
    _, rc = scn.render_count()
    rays_per_frame = rc.total()
    samples_per_frame = rc.primary

    # CORRECTNESS GATE, before anything is timed: the frame of the timed kernel variant must have the
    # fingerprint of the reference's frame (tests/golden/manifest.json).  No timing is reported otherwise.
    want_hash = golden_hash(snap)
    gate = scn.new_frame()
    scn.render(gate)
    torch.cuda.synchronize()
    frame_check = {"kind": "none: no reference frame exists for this workload", "ok": None}
    if want_hash is not None:
        got = qr.frame_hash(gate)
        frame_check = {"kind": "FNV-1a-64 of the whole frame == the reference's (tests/golden/manifest.json)",
                       "hash": f"{got:016x}", "ok": got == want_hash}
        if got != want_hash:
            print(json.dumps({"error": "rendered frame differs from the reference frame", "workload": args.workload,
                              "hash": f"{got:016x}", "expected": f"{want_hash:016x}"}), flush=True)
            raise SystemExit(2)

    ex = sharding.FrameExchange(H, W, N, rank)
    # --inflight steps are in flight (one render target set and one HIP stream each), so the long recursion
    # waves that end one launch overlap the bulk of the next instead of idling the GPU (DESIGN.md "Critical path").
    D = max(1, args.inflight)
    B = 2 * D if N > 1 else D                                           # two groups of D steps: one renders while the other is exchanged
    streams = [torch.cuda.Stream() for _ in range(D)]
    compute = streams[0]
    comm = torch.cuda.Stream()
    split = bool(args.split_frame and N > 1)
    sp = sharding.SplitFrame(H, W, N, rank) if split else None

    def split_frame_buffer():
        # height rounded up to whole tile rows (the buffers are viewed as [tile row, 8, width]); the kernel writes rows < H
        return torch.zeros((sp.alloc_rows, W), dtype=torch.int32, device=f"cuda:{local_rank}")
    if split:
        frames = [[split_frame_buffer()] for _ in range(B)]
    else:
        frames = [[scn.new_frame() for _ in range(N)] for _ in range(B)]    # render targets of the steps in flight
    finals = [split_frame_buffer() if split else scn.new_frame() for _ in range(B)]     # the frame this rank assembles
    gfinals = [[scn.new_frame() for _ in range(N)] for _ in range(B)] if (args.gather and rank == 0 and N > 1) else None
    ev_render = [torch.cuda.Event() for _ in range(B)]
    ev_comm = [torch.cuda.Event() for _ in range(2)]                    # per group of D steps
    # the N blocks this rank owns in a step (block (rank + f) mod N of frame f) go out as ONE multi-target
    # launch (qr_render_multi_async): cut into N launches the same work costs 2-2.6x (ramp, drain and tail of
    # every small grid; measured on one GPU, tools/gpu_shard_overhead.py)
    multi = [qr.MultiRender([(scn, frames[b][f]) + ex.my_rows(f) for f in range(N)]) for b in range(B)] if (N > 1 and not split) else None
    if split:
        scn.set_tile_rows(rank, N)          # every launch of this rank from here on: tile rows rank, rank + N, ...
    state = {"pending": [], "timed": False, "ev": []}

    def flush():
        """one grouped exchange for the steps rendered since the last one (fewer, larger collectives)"""
        if N > 1 and state["pending"]:
            grp = state["pending"][0] // D
            with torch.cuda.stream(comm):
                for b in state["pending"]:
                    comm.wait_event(ev_render[b])
                if state["timed"]:
                    # the exchange itself, on the stream it runs on: from the moment its inputs are rendered to its last copy
                    x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    x0.record(comm)
                if split:
                    sp.gather([(frames[b][0], finals[b]) for b in state["pending"]], root=0)
                elif args.gather:
                    ex.gather_many([(frames[b], gfinals[b] if gfinals else None) for b in state["pending"]], root=0)
                else:
                    ex.exchange_many([(frames[b], finals[b]) for b in state["pending"]])
                ev_comm[grp].record(comm)
                if state["timed"]:
                    x1.record(comm)
                    state["ev"].append((x0, x1, len(state["pending"])))
            state["pending"] = []

    def step(i):
        buf = i % B
        st = streams[i % D]
        with torch.cuda.stream(st):
            if split:
                st.wait_event(ev_comm[buf // D])
                scn.render(frames[buf][0], stream=st)
            elif N > 1:
                st.wait_event(ev_comm[buf // D])            # the exchange that last read this group of buffers is done
                multi[buf](stream=st)
            else:
                scn.render(frames[buf][0], stream=st)
            ev_render[buf].record(st)
        if N > 1:
            state["pending"].append(buf)
            if len(state["pending"]) == D:
                flush()

    # Setup, not warmup: every stream, render target, multi-target launch and (N > 1) every peer connection of both
    # exchange groups is used once here, so that a small --warmup cannot leave a lazily created stream or an RCCL
    # communicator to be set up inside the timed region (measured: --warmup 2 with three streams in flight timed the
    # first use of the third stream, 0.32 ms per step instead of 0.06).
    for i in range(B):
        step(i)
    flush()
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    flush()
    torch.cuda.synchronize()
    # How long is one pass of --steps?  A region of a millisecond (the driver's --steps 20 at 0.05 ms per step) measures the
    # ramp and the drain of the launch pipeline, not the steps: the pass is REPEATED back to back inside the one timed region
    # until the region lasts >= --min-region-ms (0.25 s), HIP events on every stream mark the boundaries between repetitions
    # (no synchronisation inside the region: the pipeline stays as full as in a long run), and ms_per_step is the MEDIAN
    # repetition.  The whole region is still bracketed by barrier + synchronize, and its own mean is reported beside it.
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    flush()
    torch.cuda.synchronize()
    probe_ms = (time.perf_counter() - t0) * 1e3
    if N > 1:
        # every rank must run the same number of passes (the exchanges are collective): agree on the slowest rank's probe
        t = torch.tensor([probe_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        probe_ms = float(t.item())
    R = 1
    if args.repetitions > 0:
        R = args.repetitions
    elif probe_ms < args.min_region_ms:
        R = int(min(100000, max(3, -(-args.min_region_ms // max(probe_ms, 1e-3)))))
    all_streams = streams + ([comm] if N > 1 else [])
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [[torch.cuda.Event(enable_timing=True) for _ in all_streams] for _ in range(R + 1)]
    t0 = time.perf_counter()
    state["timed"] = True
    for e, s_ in zip(marks[0], all_streams):
        e.record(s_)
    for r in range(R):
        for i in range(args.steps):
            step(r * args.steps + i)
        flush()
        for e, s_ in zip(marks[r + 1], all_streams):
            e.record(s_)
    state["timed"] = False
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt_region = time.perf_counter() - t0
    # boundary r = the moment the LAST stream passed its mark (the streams start idle and together: marks[0] are equal within
    # the time it takes to record them)
    bounds = [max(marks[0][0].elapsed_time(e) for e in marks[r]) for r in range(R + 1)]
    # Short passes (the driver's --steps 20 = 0.85 ms) do not end on the same stream each time (20 steps over 3 streams), so single
    # passes alternate between two lengths; the median is taken over GROUPS of consecutive passes of >= 20 ms each (at least 3
    # groups, at most 15): per-step time of a group = its span / its steps.
    per_group = max(1, min(R // 3 if R >= 3 else 1, int(-(-20.0 // max(probe_ms, 1e-3)))))
    n_groups = R // per_group
    grp_ms = sorted((bounds[(g + 1) * per_group] - bounds[g * per_group]) / per_group for g in range(n_groups))
    rep_ms = sorted(bounds[r + 1] - bounds[r] for r in range(R))
    if n_groups >= 3:
        dt = grp_ms[n_groups // 2] * 1e-3             # the median group, as the time of ONE pass of --steps steps
    else:
        dt = dt_region / R                            # too few passes for a median: the region's own mean
    if N > 1:
        t = torch.tensor([dt, dt_region], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_region = float(t[0].item()), float(t[1].item())
    total_steps = R * args.steps
    timing = {"repetitions": R, "steps_per_repetition": args.steps, "timed_region_ms": dt_region * 1e3,
              "ms_per_step_region_mean": dt_region / total_steps * 1e3,
              "ms_per_step_median_group": dt / args.steps * 1e3,
              "passes_per_group": per_group, "groups": n_groups,
              "ms_per_step_min_group": grp_ms[0] / args.steps, "ms_per_step_max_group": grp_ms[-1] / args.steps,
              "ms_per_step_min_rep": rep_ms[0] / args.steps, "ms_per_step_max_rep": rep_ms[-1] / args.steps,
              "how": "one timed region (barrier + synchronize on both sides) of `repetitions` back-to-back passes of `steps` steps; "
                     "HIP events on every launch stream mark the pass boundaries; ms_per_step = median over groups of consecutive passes "
                     "(>= 20 ms each) of group span / group steps" + ("" if n_groups >= 3 else " (fewer than 3 groups: the region's mean)")}

    # the frames the TIMED steps produced are checked too: every buffer still in flight at the end
    ok = True
    if N > 1:
        # the assembled frame must equal a whole-frame render of this rank (which passed the gate above)
        last = (total_steps - 1) % B
        if split:
            ok = bool((gate == finals[last][:H]).all().item()) if rank == 0 else True
        elif args.gather:
            ok = all(bool((gate == g).all().item()) for g in gfinals[last]) if rank == 0 else True
        else:
            ok = bool((gate == finals[last]).all().item())
        flag = torch.tensor([1 if ok else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    else:
        for b in range(min(B, total_steps + args.warmup)):
            ok = ok and bool((gate == frames[b][0]).all().item())
    frame_check["timed_frames_match"] = ok
    if not ok:
        if rank == 0:
            print(json.dumps({"error": "a frame produced inside the timed region differs from the checked frame",
                              "workload": args.workload}), flush=True)
        raise SystemExit(3)
    collective = None
    if N > 1:
        devs = [None] * N
        dist.all_gather_object(devs, f"rank {rank}: cuda:{torch.cuda.current_device()} {torch.cuda.get_device_name()}")
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "devices": devs,
                      "pattern": ("one frame's tile rows dealt round-robin, gathered on rank 0 by grouped point-to-point sends" if split else
                                  "grouped point-to-point gather of row blocks to rank 0" if args.gather else
                                  "grouped point-to-point all-to-all of row blocks") + ", one call per step group (sharding.py)"}
        if state["ev"]:
            # per-step exchange time of THIS rank, HIP events on the communication stream inside the timed region (with the
            # gloo backend the stream only sees the staging copies; the host part is in ms_per_step)
            ms = [a.elapsed_time(b) for a, b, _ in state["ev"]]
            steps_x = sum(n for _, _, n in state["ev"])
            collective.update(exchange_groups=len(ms), exchange_ms_per_group=sum(ms) / len(ms), exchange_ms_per_step=sum(ms) / max(1, steps_x),
                              exchange_ms_max_group=max(ms))

    # dominant-kernel duration: HIP events recorded on the launch stream around full-frame launches
    scn.set_rows(0, H, 0, 1)
    scn.set_tile_rows(0, 1)
    avg_ms, min_ms = scn.render_timed(frames[0][0], max(20, min(args.steps, 200)), stream=compute)
    torch.cuda.synchronize()

    if rank == 0:
        frames_done = args.steps * (1 if split else N)  # per pass of --steps; dt is the median pass
        total_rays = rays_per_frame * frames_done
        value = total_rays / dt / 1e6
        cpu = None
        if N == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload, rays_per_frame, gpu_frame=gate.cpu().numpy().view("uint32"))
            band = cpu.get("band") if cpu else None
            if band is not None and "gpu_pixels_differ" in band:
                frame_check = {"kind": f"rows {band['rows'][0]}..{band['rows'][1]} equal the oracle's pixels (no reference frame exists for a synthetic scene)",
                               "ok": band["gpu_pixels_differ"] == 0, "timed_frames_match": ok}
                if band["gpu_pixels_differ"] != 0:
                    print(json.dumps({"error": "rendered rows differ from the oracle", "workload": args.workload}), flush=True)
                    raise SystemExit(2)
        # Which roofline: the path is scalar-per-ray fp32 VALU work -> the VECTOR ALU is the primary
        # bound; peak = the FMA-counting fp32 vector peak (bit-exactness forbids contraction, so half of it is the most
        # un-fused arithmetic can reach).  TWO counts of the algorithmic fp32 operations of a frame exist, both with the
        # weights of SURVEY 8(d) (div and sqrt = 1), and EVERY workload's line carries both with both fractions:
        #   flops_oracle  the reference's algorithm as the CPU restatement executes it: every element of every list
        #                 walked, no cull (tests/golden/work.json "deferred"; for a scene the oracle cannot walk whole --
        #                 config 5 -- its count on a band of rows scaled by rays: "oracle_band", or this run's cpu_baseline band)
        #   flops_kernel  the same weights added by every lane of the kernel for each step it actually executes
        #                 (QR_PROF build, tools/gpu_work.py: work.json "kernel"); the culls' own arithmetic is not in it
        # `achieved` / `frac` use the oracle's count where it covers the whole frame (SURVEY 8(d) defines the figure by the CPU
        # restatement's counters), else the kernel's, and `basis` says which; achieved_oracle / frac_oracle and achieved_kernel /
        # frac_kernel always stand beside them.  All over the kernel's mean launch duration measured here with HIP events.
        work_all = all_work(snap)
        flops_oracle, oracle_src, flops_kernel, kernel_src = None, None, None, None
        d = work_all.get("deferred")
        if d is not None and d.get("rays") == rays_per_frame and "kernel-side" not in d.get("source", ""):
            flops_oracle, oracle_src = d["flops"], "oracle count of the whole frame, tests/golden/work.json"
        elif cpu is not None and cpu.get("band"):
            flops_oracle = int(cpu["band"]["flops"] / max(1, cpu["band"]["rays"]) * rays_per_frame)
            oracle_src = (f"oracle count on rows {cpu['band']['rows'][0]}..{cpu['band']['rows'][1]} of this run's cpu_baseline, scaled by rays "
                          "(the oracle walks the LISTS of this scene; the kernel walks grids)")
        elif work_all.get("oracle_band") is not None:
            ob = work_all["oracle_band"]
            flops_oracle = int(ob["flops"] / max(1, ob["rays"]) * rays_per_frame)
            oracle_src = f"oracle count on rows {ob['rows'][0]}..{ob['rows'][1]} (tests/golden/work.json oracle_band), scaled by rays"
        k = work_all.get("kernel")
        if k is not None and k.get("rays") == rays_per_frame:
            flops_kernel, kernel_src = k["flops"], k.get("source", "kernel-side count") + ", tests/golden/work.json"
        counters = committed_counters(args.workload)
        if counters is not None and counters.get("stale"):
            roofline_stale, counters = counters, None
        else:
            roofline_stale = None
        roofline = dict(bound="valu", achieved=None, peak=VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=None,
                        basis="flops_oracle" if flops_oracle is not None else ("flops_kernel" if flops_kernel is not None else None),
                        traffic=(counters or {}).get("hbm_bytes_per_launch"),
                        kernel=qr.lib().qr_kernel_name().decode(), kernel_avg_ms=avg_ms, kernel_min_ms=min_ms,
                        flops_oracle=flops_oracle, flops_oracle_source=oracle_src,
                        flops_kernel=flops_kernel, flops_kernel_source=kernel_src)
        if roofline_stale is not None:
            roofline["counters"] = roofline_stale
        for name, fl in (("oracle", flops_oracle), ("kernel", flops_kernel)):
            if fl is not None:
                tf = fl / (avg_ms * 1e-3) / 1e12
                roofline[f"achieved_{name}"] = tf
                roofline[f"frac_{name}"] = tf / VALU_PEAK_TFLOPS
        # headline basis: the oracle's count where it counted the WHOLE frame (every reference scene); where it only has a band of
        # rows to scale (config 5: its list walk costs 6 000 operations a ray, the kernel's grid walk 26), the kernel's own count
        whole = flops_oracle is not None and oracle_src.startswith("oracle count of the whole frame")
        flops = flops_oracle if (whole or flops_kernel is None) else flops_kernel
        roofline["basis"] = None if flops is None else ("flops_oracle" if flops is flops_oracle else "flops_kernel")
        if flops is not None:
            tf = flops / (avg_ms * 1e-3) / 1e12
            roofline.update(achieved=tf, frac=tf / VALU_PEAK_TFLOPS, flops_per_launch=flops)
        if counters is not None and counters.get("thread_cycles_valu") and flops_kernel:
            # how many vector lane-operations the kernel issues per algorithmic operation it executes (SQ_THREAD_CYCLES_VALU of
            # the committed counter pass of this build / flops_kernel): the overhead factor of the implementation
            roofline["valu_lane_ops_per_flop"] = counters["thread_cycles_valu"] / flops_kernel
            if flops_oracle:
                roofline["valu_lane_ops_per_flop_oracle"] = counters["thread_cycles_valu"] / flops_oracle
        # the HBM leg, for the record (deliberate correction of SURVEY 8(d): the kernel reads the scene through the
        # scalar cache, not once per workgroup, so the algorithmic bytes are one frame write + one scene read)
        alg_bytes = 4 * W * H + int(scn.info.device_bytes)
        gbs = alg_bytes / (avg_ms * 1e-3) / 1e9
        roofline["hbm"] = dict(achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                               algorithmic_bytes_per_launch=alg_bytes)
        if counters is not None:
            roofline["counters"] = counters
            # What actually bounds the kernel: instruction issue.  MI355X_MICROARCH.md: a CDNA4 SIMD is 32 lanes wide and issues a
            # wave64 VALU instruction every 2 cycles = 1.2 G wave-instructions/s per SIMD at 2.4 GHz, 1024 SIMDs (peak).  What a
            # SIMD sustains at the renderer's occupancy was measured with tools/ubench/issue_rate.hip (profiles/r04_valu_issue_rate.txt):
            # 0.94 G/s at 4 waves per SIMD (0.89 at 2, 1.03 at 8; the chip runs at 2.15-2.3 GHz under that load; SALU instructions
            # share the slot) -- `measured_ceiling`.  The instruction count is the committed counter value (not measured in this
            # run), the time is this run's: achieved = issued per launch / kernel time; with several launches in flight the
            # per-frame rate is the one that counts.
            insts = sum(counters.get(k, 0.0) for k in ("insts_valu", "insts_salu", "insts_smem", "insts_vmem_rd", "insts_vmem_wr"))
            if insts > 0:
                peak = 1.2 * 1024
                ceiling = 0.94 * 1024
                per_launch = insts / (avg_ms * 1e-3) / 1e9
                per_frame = insts * args.steps / dt / 1e9       # one frame's worth of instructions per GPU and step (median pass)
                roofline["issue"] = dict(unit="G wave-instructions/s", peak=peak, measured_ceiling=ceiling, wave_instructions_per_launch=insts,
                                         achieved_isolated_launch=per_launch, frac_isolated_launch=per_launch / peak,
                                         achieved_in_flight=per_frame, frac_in_flight=per_frame / peak,
                                         frac_in_flight_of_measured_ceiling=per_frame / ceiling,
                                         source="MI355X_MICROARCH.md (peak: SIMD-32, 2 cycles per wave64 VALU), profiles/r04_valu_issue_rate.txt "
                                                "(measured ceiling), profiles/counters.json (count), this run (time)",
                                         measured_in_this_run=False)
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": value, "unit": "Mrays/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic scene (quadray-engine_amd/synth.py)" if snap.startswith("synth:") else
                     f"the reference's own scene: snapshot of the engine's scene graph (tests/golden/{snap}.qrs.gz), no dataset involved"),
            "config": {"workload": f"{args.workload}: {desc}", "resolution": [W, H],
                       "frames_per_step": 1 if split else N, "steps_in_flight": D, "rays_per_frame": rays_per_frame,
                       "rays": rc.as_dict(),
                       "parallelism": (f"one frame per step, tile rows round-robin over {N} ranks, gathered on rank 0 per {D} steps" if split else
                                       f"tile-row blocks x{N}, 1 multi-target launch/step, 1 grouped exchange per {D} steps") if N > 1 else "single GPU",
                       "fps": frames_done / dt, "msamples_per_s": samples_per_frame * frames_done / dt / 1e6,
                       "frame_check": frame_check, "assembled_frame_matches": ok},
            "timing": timing,
            "lib": qr.lib().qr_version().decode(),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if collective is not None:
            out["collective"] = collective
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
