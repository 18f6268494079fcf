#!/usr/bin/env python3
"""bench.py — Mrays/s (fwd+bwd) of the K-shell render hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched by
torch.distributed.run, one rank per GPU).  Rank 0 prints ONE JSON line.

A step = one forward + backward pass of the hot path over one 800x800 frame of
synthetic rays (BASELINE.json configs[1]: K=5 shells, neural-texture appearance),
inputs resident in HBM before the timed region.  Rays shard across ranks by
16 384-ray chunks, round-robin (tile-parallel, no data-path collective in the
render) -> "scaling": "weak" is reported with a full frame PER RANK.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from volsurfs_amd.pipeline import GRAD_CHAIN_GAIN  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec)
MFMA_F16_PEAK_TFLOPS = 2500.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 200 (frame) / 500 (train*) / 3 frames (dtu)")
    ap.add_argument("--warmup", type=int, default=None, help="default 10 (frame, dtu) / 100 (train*)")
    ap.add_argument("--workload", default="frame", choices=["frame", "train", "train-permuto", "dtu", "render"],
                    help="frame (default): the headline metric, fwd+bwd of one 800x800 frame.  train: the "
                         "reference's training loop (TensorReel batches, dynamic ray count -> 49 152 hits, "
                         "fwd+bwd+Adam) on the neural-texture appearance.  train-permuto: BASELINE configs[2] "
                         "as written (legacy permutohash + MLP appearance).  dtu: configs[3]'s learned "
                         "background (NerfHash, 32 samples per ray) fwd+bwd+Adam on 65 536-ray batches.  render: forward-only "
                         "full-frame evaluation (BaseMethod.render), live neural textures and baked 8-bit textures")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1, frame workload: weak = one full frame per rank (default); strong = ONE frame "
                         "dealt to the ranks in 8-row bands, round-robin (parallel.shard_bands)")
    ap.add_argument("--grad-wire-dtype", default="f32", choices=["f32", "bf16"],
                    help="N > 1: width of the gradient all-reduce on the wire (bf16 halves the bytes, rounds the sum)")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1, train workloads: reduce-scatter the gradients, Adam on each rank's slice of every "
                         "tensor, all-gather the f16 copies (optim.ShardedFusedAdam) instead of all-reduce + full Adam")
    ap.add_argument("--adam-overlap", type=int, default=0, choices=[0, 1],
                    help="train workload (neural textures, one rank): the Adam launch on a side stream as a bounded grid "
                         "(vsa_adam_step_shared) beside the next iteration's ray batch, traversal and texel compaction")
    ap.add_argument("--target-hits", type=int, default=49152)
    ap.add_argument("--views", type=int, default=50)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--width", type=int, default=None,
                    help="frame workload: frame width when it is not square (--res is then the height), e.g. "
                         "--res 1080 --width 1920 --shells 7 --subdiv 7 = BASELINE configs[4]'s per-GPU frame")
    ap.add_argument("--shells", type=int, default=5)
    ap.add_argument("--subdiv", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--noise", type=float, default=0.0, help="frame workload: radial noise of the shells (0: perfect spheres)")
    ap.add_argument("--atlas-charts", type=int, default=0,
                    help="frame workload: cut every shell's uv parameterisation into G x G randomly packed charts")
    ap.add_argument("--init", default="tcnn", choices=["tcnn", "spread"],
                    help="frame workload: parameter initialisation (tcnn: U(+-1e-4) tables; spread: U(+-1))")
    ap.add_argument("--no-noisy", action="store_true",
                    help="skip the extra scenes the default frame line reports beside the headline: value_noisy "
                         "(noise 0.05, 6x6 uv charts, spread parameters), value_stress (mesh.stress_shells: "
                         "non-convex lobed shells, 12x triangle-area spread, 256 charts) and value_cold (the "
                         "headline scene with NO inter-frame feedback and a camera that moves every step)")
    ap.add_argument("--stress", action="store_true",
                    help="frame workload: run the stress scene (mesh.stress_shells) as the main scene")
    ap.add_argument("--cold", action="store_true",
                    help="frame workload: no inter-frame feedback (stateless traversal launch order, fitted work "
                         "split) and a camera that orbits --orbit-deg per step")
    ap.add_argument("--orbit-deg", type=float, default=5.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1: still form a one-rank process group (--dist-backend, default nccl = RCCL) and "
                         "drive the data-parallel schedule through it: gradient all-reduce per shell during "
                         "backward, as the N > 1 ranks do")
    ap.add_argument("--no-graph", action="store_true", help="time the eager path instead of the HIP-graph replay")
    ap.add_argument("--by-shell", action="store_true",
                    help="1 GPU: run the data-parallel step (phased hash-grid backward + device flags) "
                         "without the collectives, to price it")
    ap.add_argument("--train-graph", type=int, default=1,
                    help="train workload (neural textures, one rank): 1 (default) = also run the iteration as one replayed HIP "
                         "graph (trainer.GraphTrainLoop) and report it as `value` beside `value_eager`; 0 = the eager loop only")
    ap.add_argument("--dp-phases-auto", type=int, default=1,
                    help="N > 1 (or --force-dist): 1 (default) = choose the phase split from the all-reduce time, the hash-grid "
                         "backward and the step's head measured at start-up (parallel.choose_phases); 0 = the fixed default")
    ap.add_argument("--dp-phases", default=None,
                    help="data-parallel step: shell-range ends of the hash-grid backward's phases, e.g. 3,5 "
                         "(default: parallel.default_phases — 3 phases)")
    ap.add_argument("--dp-wait", type=int, default=0, choices=[0, 1, 2],
                    help="how the communication stream waits for a device flag: 1 hipStreamWaitValue32, "
                         "2 a one-lane polling kernel, 0 the first where the device offers it")
    ap.add_argument("--dp-reserve-cus", type=int, default=None,
                    help="data-parallel step: compute units the hash-grid backward leaves free for the collectives' kernels "
                         "(default: 16 when more than one rank runs — whether RCCL's kernels fit BESIDE a workgroup that "
                         "holds most of a CU's registers cannot be measured on one GPU, 16 whole CUs cost the launch 6 %% — "
                         "else 0)")
    ap.add_argument("--dp-comm", default="auto", choices=["auto", "rccl", "torch"],
                    help="data-parallel step: all-reduces through RCCL called directly on the side stream (rccl), through "
                         "torch.distributed (torch), or rccl when the backend is nccl (auto)")
    ap.add_argument("--dp-split-launches", action="store_true",
                    help="round 4's data-parallel schedule (K hash-grid backward launches, eager), for A/B")
    ap.add_argument("--cpu-sample-rays", type=int, default=16384,
                    help="rays of the cpu_baseline sample: one evaluation chunk of the reference "
                         "(test_rays_batch_size = 16 384, config/volsurfs/base_5.cfg:8; SURVEY 8d)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="functional tests of the multi-rank path on one GPU: gloo + --single-device")
    ap.add_argument("--single-device", action="store_true",
                    help="all ranks use cuda:0 (testing only; RCCL refuses two ranks on one GPU)")
    args = ap.parse_args(argv)
    training = args.workload.startswith("train")
    if args.steps is None:
        # frame: 200 steps = 0.64 s of back-to-back graph replays (the round-1 default of 20 was a 63 ms
        # timed region, too short for any external GPU-busy sampler to corroborate)
        # (render: 200 frames = 0.2 s — at 50 the eager loop's 50 ms timed region moved by +-10 % from run to run on one box)
        args.steps = 500 if training else (3 if args.workload == "dtu" else 200)
    if args.warmup is None:
        args.warmup = 100 if training else 10
    return args


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is
    ONE JSON line there.  File descriptor 1 is pointed at stderr while the process group forms."""

    def __enter__(self):
        sys.stdout.flush()
        self._keep = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)      # the banner sits in libc's stdio buffer (a pipe is fully buffered)
        os.dup2(self._keep, 1)
        os.close(self._keep)


def dist_info(dist, args):
    """What the line's n_gpus rests on: the size of the process group the ranks really formed."""
    if dist is None:
        return {"ranks_seen": 1, "dist_backend": None}
    return {"ranks_seen": dist.get_world_size(), "dist_backend": dist.get_backend()}


FRAME_PATH_SOURCES = ("common.h", "nt_common.h", "nt_enc_common.h", "nt_mlp_common.h", "nt_quant_table.h",
                      "nt_texels.hip", "nt_encode.hip", "nt_mlp.hip", "nt_fused.hip", "nt_shade.hip", "trace.hip",
                      "composite_dense.hip", "raygen.hip")


def kernel_source_hash():
    """sha256 over the sources of the kernels the frame workload launches (the ones
    profiles/traffic.json holds PMC traffic for)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "volsurfs_amd", "csrc")
    for n in FRAME_PATH_SOURCES:
        h.update(n.encode())
        h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()


def workload_key(args):
    """What a PMC traffic collection was taken ON: profiles/traffic.json is valid for the kernel
    sources (kernel_source_hash) AND this workload only — another frame size, shell count or scene
    moves other bytes."""
    return (f"frame {args.width or args.res}x{args.res} K={args.shells} subdiv={args.subdiv} noise={args.noise} "
            f"charts={args.atlas_charts} init={args.init} stress={int(bool(args.stress))} cold={int(bool(args.cold))}")


def pmc_busy(args, kernel):
    """PMC-derived pipe occupancy of a kernel on this workload (profiles/pmc_summary.json, written by
    tools/make_pmc_json.py from separate rocprofv3 --pmc passes): {"mfma_busy", "valu_insts", ...}, or None when
    the file holds nothing for this workload at these kernel sources."""
    f = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(f):
        return None
    e = json.load(open(f)).get("workloads", {}).get(workload_key(args))
    if not e or e.get("_kernel_source_sha256") != kernel_source_hash():
        return None
    return e.get(kernel)


def trace_ceiling(pipe, trace_ms, args, nr_cus=256):
    """Where the traversal launch stands against what bounds it.  HBM bandwidth is the wrong
    yardstick (5 % of it, traffic = algorithmic bytes).  Counted with vsa_trace_q_stats (the same walk with
    counters): lane-level node visits and triangle tests, wave-level trips (a trip = one node fetch + two slab tests
    + the stack step of a whole wave).  With the PMC passes of profiles/pmc_summary.json: 95 % of the node fetches hit
    the vector L1, the L1->L2 round trip of the rest is ~275 cycles, and the vector pipe is 71 % busy at a lane
    utilisation of 0.50 — the launch is bound by VECTOR ISSUE UNDER DIVERGENCE (a trip costs a wave ~40 instructions
    whatever the number of live lanes), with 1.4x of headroom to a saturated pipe, not by any memory level."""
    ro, rd = (pipe._o_t, pipe._d_t) if pipe.image_hw is not None else (pipe.rays_o, pipe.rays_d)
    st = pipe.tracer.walk_stats(ro, rd)
    sec = trace_ms * 1e-3
    l2_bytes = st["lane_visits"] * 32 + st["tri_tests"] * 48
    out = {
        "lane_node_visits": st["lane_visits"], "lane_tri_tests": st["tri_tests"], "wave_trips": st["wave_trips"],
        "waves": st["waves"], "max_wave_trips": st["max_wave_trips"],
        "lane_utilisation": st["lane_visits"] / max(1.0, 64.0 * st["wave_trips"]),
        "Gvisits/s": st["lane_visits"] / sec / 1e9, "Gtri_tests/s": st["tri_tests"] / sec / 1e9,
        "requested_GB/s": l2_bytes / sec / 1e9, "wave_trips/s": st["wave_trips"] / sec,
        "ns_per_trip_and_resident_wave": 1e9 * sec * nr_cus * 10 / max(1, st["wave_trips"]),
        "bound": "vector issue under divergence"}
    pm = pmc_busy(args, "trace")
    if pm:
        out["pmc"] = {k: v for k, v in pm.items() if k != "counters_per_launch"}
        if "valu_issue_busy" in pm:
            out["frac_of_vector_issue"] = pm["valu_issue_busy"]
    return out


def kernel_table(fn, iters):
    """Per C-ABI entry point: ms per iteration (median call x calls per iteration) over `iters` calls of fn(), events
    directly around every library launch (the torch glue in between is not in it)."""
    from volsurfs_amd import _lib
    _lib.kernel_events = {}
    for _ in range(iters):
        fn()
    tot, med = _lib.kernel_totals(), _lib.kernel_ms()
    _lib.kernel_events = None
    # median per call x calls per iteration: over three iterations ONE call that carried a first-use cost (a lazily
    # built descriptor, a rebalance) owned the mean — r5's training line quoted nt_mlp_bwd at 0.21 ms where the
    # workgroups' own clocks and rocprof say 0.09-0.105 (profiles/r06/mlp_bwd_train_stamps.txt)
    return {k: {"ms_per_iter": round(med[k] * n / iters, 4), "calls_per_iter": round(n / iters, 2)}
            for k, (ms, n) in sorted(tot.items(), key=lambda kv: -med[kv[0]] * kv[1][1])}


def cpu_baseline(pipe, sample_rays):
    """The oracle (oracle/pipeline.py: CPU restatement of the reference path,
    evaluated the reference's way — 4 network evaluations per hit, degree and
    model) timed on this box's host cores over a bounded sample of the same
    workload, forward + backward.  kind="port": the reference has no CPU path of
    its own.

    The oracle's result is not thrown away (VERDICT r5 missing #4): the SAME sample of the full-size frame —
    full-size meshes, full-resolution textures, the frame's parameters — also goes through the HIP path (a
    second pipeline over the sampled rays that shares the BVH; outside every timed region) and the two are
    compared: hits, per-shell values, RGB, every gradient (oracle/parity.py).  Returns (cpu_baseline, parity_sample)."""
    from oracle import parity as opar
    from volsurfs_amd.pipeline import KShellPipeline
    # the reference's driver scripts assume 16 host threads (scripts/volsurfs.sh:47);
    # many more than that makes torch-CPU's scatter backward crawl on big hosts
    torch.set_num_threads(min(16, os.cpu_count()))
    n = min(sample_rays, pipe.nr_rays)
    idx = torch.linspace(0, pipe.nr_rays - 1, n, device=pipe.rays_o.device).long()
    fb, pipe.tracer._fb = pipe.tracer._fb, None            # (the launch-order feedback belongs to the frame's rays)
    # (parameters: the frame's MLP weights with SPREAD hash tables, U(-1, 1) — with the tcnn initialisation the frame is
    #  timed at, U(+-1e-4), every texel is sigmoid(~0) and the oracle's own fp16 autograd is mostly rounding noise: a
    #  comparison that could not fail on RGB and could not pass on gradients)
    sub = KShellPipeline(pipe.meshes, pipe.rays_o[idx].contiguous(), pipe.rays_d[idx].contiguous(),
                         pipe.gt[idx].contiguous(), tracer=pipe.tracer, init="spread", seed=5)
    with torch.no_grad():
        sub.bank.weights.copy_(pipe.bank.weights)
    sub.bank.refresh_half_params()
    rgb = sub.step()
    torch.cuda.synchronize()
    ref, dt = opar.oracle_step(sub, loss_scale=128.0)      # the reference's fp16 autograd runs under tcnn's loss scale
    parity = opar.compare_step(sub, rgb, ref)
    parity["sample"] = (f"{n} rays spread over the frame (every {pipe.nr_rays // n}-th), L1 mean over the sample, full-size "
                        "meshes and textures, hash tables U(-1, 1), the frame's MLP weights; HIP path vs oracle.pipeline.render_step")
    pipe.tracer._fb = fb
    base = {"value": n / dt / 1e6, "unit": "Mrays/s", "cores": torch.get_num_threads(), "kind": "port",
            "frame_extrapolation_s": dt * pipe.nr_rays / n,
            "sample": f"{n} rays (one evaluation chunk of the reference) spread over the same frame, K={pipe.K}: "
                      f"brute-force closest hit (oracle/raytrace_ref.c, OpenMP over the host's cores) + per-hit SH "
                      f"neural textures fwd+bwd (oracle/neural_texture.py on torch-CPU, {torch.get_num_threads()} "
                      f"threads) + composite fwd+bwd (oracle/composite.py); {dt:.1f} s"}
    return base, parity


def synthetic_reel(n_views, res, device, seed=42):
    """`n_views` cameras on a sphere of radius 1.5 around the shells (NeRF-synthetic intrinsics),
    uniform-random ground-truth images: the data a `TensorReel` holds in HBM (trainer.py:176-190)."""
    from volsurfs_amd.camera import Camera, TensorReel
    g = torch.Generator().manual_seed(seed)
    cams = []
    for i in range(n_views):
        z = 1.0 - 2.0 * (i + 0.5) / n_views                   # spiral over the sphere
        r, phi = (1.0 - z * z) ** 0.5, i * 2.399963
        eye = (1.5 * r * float(np.cos(phi)), 1.5 * z, 1.5 * r * float(np.sin(phi)))
        cams.append(Camera.look_at(eye, focal=1111.1 * res / 800.0, height=res, width=res, device=device))
    rgbs = torch.rand(n_views, res, res, 3, generator=g)
    return TensorReel(cams, rgbs, device=device)


def run_train(args, world, rank, dev, dist):
    """The reference's training loop (trainer.py:110-308) at config-3 shape: batches drawn from a
    TensorReel, dynamic ray count steering the hit count to --target-hits, forward + L1 + backward
    + fused Adam + scheduler per iteration; `--warmup` untimed iterations, then `--steps` timed."""
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.trainer import train_step_from_reel
    legacy = args.workload == "train-permuto"
    meshes = nested_shells(K=args.shells, subdiv=args.subdiv, device=dev)
    max_rays = 1 << 17
    kw = dict(using_neural_textures=False, rgb_pos_encoder_type="permutohash",
              rgb_mlp_layers_dims=(128, 128, 64)) if legacy else {}
    # data-parallel replicas are ONE model: every rank builds it from the same seeds (the legacy
    # MLPs / encoders draw from the global torch RNG); only the data (reel) differs per rank
    torch.manual_seed(42)
    method = VolSurfs(meshes, max_rays=max_rays, nr_warmup_iters=500, seed=42, **kw)
    method.init_optim(world=world, rank=rank, sharded=args.sharded_adam)
    torch.manual_seed(42 + rank)
    reel = synthetic_reel(args.views, args.res, dev, seed=42 + rank)
    target = args.target_hits // world               # each rank draws its share of the global batch
    state = {"nr_rays": 512, "it": 0, "rays": 0, "hits": 0}

    def one(count):
        n = state["nr_rays"]
        method.grad_scale = GRAD_CHAIN_GAIN * float(n)     # mean-L1 over n rays: no device read-back for the scale
        losses, nxt = train_step_from_reel(method, reel, n, jitter_pixels=True, iter_nr=state["it"],
                                           is_first_iter=state["it"] == 0,
                                           target_nr_of_training_samples=target, world=world,
                                           sync_losses=False, overlap_optimizer=bool(args.adam_overlap) and world == 1)
        if count:
            state["rays"] += n
            state["hits"] += int(getattr(method, "last_nr_samples", 0))
        state["nr_rays"] = max(64, min(int(nxt), 4 * max_rays))
        state["it"] += 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        one(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one(True)
    barrier()
    dt = time.perf_counter() - t0
    # r6: the same loop as ONE replayed HIP graph per iteration (trainer.GraphTrainLoop: every launch at a fixed capacity
    # of rays, the dynamic ray count / schedule / Adam step count / sampler stream advanced on the device) — the eager
    # loop's host side and launch gaps are as long as its kernels.  Same batches, same rays, same learning rates.
    graph = None
    if not legacy and world == 1 and args.train_graph:
        from volsurfs_amd.trainer import GraphTrainLoop
        method.grad_scale = None
        loop = GraphTrainLoop(method, reel, state["nr_rays"], target, iter_nr=state["it"]).capture()
        for _ in range(20):
            loop.step()
        barrier()
        s0 = loop.read()
        tg = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        barrier()
        dtg = time.perf_counter() - tg
        s1 = loop.finish()
        state["it"], state["nr_rays"] = int(s1["iter"]), int(s1["nr_rays"])
        graph = {"it/s": args.steps / dtg, "ms_per_step": dtg / args.steps * 1e3,
                 "rays_per_iter": (s1["sum_rays"] - s0["sum_rays"]) / args.steps,
                 "hits_per_iter": (s1["sum_hits"] - s0["sum_hits"]) / args.steps,
                 "capacity": loop.capacity, "clamped": int(s1["clamped"]), "loss": float(s1["loss"])}
    # fixed cost per iteration: the same loop on batches of 64 rays (launches, host syncs, the
    # mark / compact scan of the texel domains, LDS staging of the level tables, Adam over all
    # parameters) — what does not shrink with the batch
    keep = state["nr_rays"]
    barrier()
    t1 = time.perf_counter()
    for _ in range(50):
        state["nr_rays"] = 64
        one(False)
    barrier()
    fixed_ms = (time.perf_counter() - t1) / 50 * 1e3
    state["nr_rays"] = keep
    in_sync = None
    if dist is not None:
        t = torch.tensor([dt, float(state["rays"]), float(state["hits"])], device=dev, dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        dt, state["rays"], state["hits"] = t[0].item(), t[1].item(), t[2].item()
        # the replicas must still be one model after the run: same parameters bit for bit
        method.sync_params()
        ck = torch.stack([p.detach().double().sum() for g in method.optimizer.param_groups for p in g["params"]])
        hi, lo = ck.clone(), ck.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        in_sync = bool((hi == lo).all().item())
    ktab = roof = None
    if rank == 0 and dist is None:
        ktab = kernel_table(lambda: one(False), 20)
        top = next(iter(ktab))
        roof = {"kernel": top, "kernel_ms": ktab[top]["ms_per_iter"] / max(1.0, ktab[top]["calls_per_iter"])}
        if legacy and top.startswith("vsa_mlp_"):
            # the fp32 matrix-core MLPs of the legacy appearance models (models/rgb.py:139: 66 -> 128 -> 128 -> 64 -> C):
            # executed FLOPs of all K shells' networks over the iteration's hits (a frac for configs[2])
            hits = int(getattr(method, "last_nr_samples", 0))
            fl_fwd = 0
            for typ in ("rgb", "alpha"):
                mods = [mm for k_, mm in method.models.items() if k_.split("_")[0] == typ and mm is not None]
                if mods:
                    dims = [[l.in_features, l.out_features] for l in mods[0].mlp.layers if isinstance(l, torch.nn.Linear)]
                    fl_fwd += 2 * sum(a * b for a, b in dims) * hits
            mult = 2 if "bwd" in top else 1
            sec = ktab[top]["ms_per_iter"] * 1e-3
            roof.update(bound="mfma", achieved=fl_fwd * mult / sec / 1e12, peak=157.3, unit="TFLOP/s",
                        frac=fl_fwd * mult / sec / 1e12 / 157.3, traffic=None,
                        units={"hits": hits, "calls_per_iter": ktab[top]["calls_per_iter"], "dtype": "f32 (v_mfma_f32_32x32x2_f32)"})
        if not legacy:        # algorithmic bytes of the neural-texture stages at THIS batch (neural_textures.stage_accounting)
            from volsurfs_amd.neural_textures import stage_accounting
            n_last = state["nr_rays"]
            hits = int(getattr(method, "last_nr_samples", 0))
            acct, fl, slots = stage_accounting(method.bank, n_last, hits)
            key = top.replace("vsa_", "").replace("_phased", "")
            if key in acct:
                sec = roof["kernel_ms"] * 1e-3
                if key in ("nt_mlp_fwd", "nt_mlp_bwd"):
                    f = fl * (2 if key.endswith("bwd") else 1)
                    roof.update(bound="mfma", achieved=f / sec / 1e12, peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                                frac=f / sec / 1e12 / MFMA_F16_PEAK_TFLOPS, traffic=None)
                else:
                    roof.update(bound="hbm", achieved=acct[key] / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=acct[key] / sec / 1e9 / HBM_PEAK_GBS, traffic=None)
                roof["units"] = {"rays": n_last, "hits": hits, "unique_texels": slots}
                # (events around ONE launch of a host-driven loop read 0.16-0.18 ms for this kernel where rocprofv3's
                #  average over the same loop and the workgroups' own clocks read 0.09-0.105: profiles/r06/
                #  kernel_stats_train.csv, mlp_bwd_train_stamps.txt — the frame line's kernel_ms, taken on 11 M pairs, agrees
                #  with rocprofv3)
                roof["kernel_ms_note"] = ("events around a single launch in the host-driven loop; rocprofv3 averages the same "
                                          "kernel at 0.09-0.105 ms at this batch (profiles/r06/kernel_stats_train.csv)")
    if rank == 0:
        ms = dt / args.steps * 1e3
        nparams = sum(p.numel() for g in method.optimizer.param_groups for p in g["params"])
        out = {
            "metric": "training iterations/s (fwd+bwd+Adam, dynamic batch)",
            # the graph loop when it ran (neural textures, one rank): the same iterations, replayed; the eager loop beside it
            "value": graph["it/s"] if graph else args.steps / dt, "value_eager": args.steps / dt,
            "train_graph": graph,
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": graph["ms_per_step"] if graph else ms, "ms_per_step_eager": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 master / f16 compute" if not legacy else "f32", "data": "synthetic",
            "Mrays/s": state["rays"] / dt / 1e6, "Mhits/s": state["hits"] / dt / 1e6,
            "rays_per_iter": state["rays"] / args.steps, "hits_per_iter": state["hits"] / args.steps,
            # (capped: the 64-ray loop forfeits the traversal look-ahead, whose prefetch is sized for the
            #  dynamic ray count, so on a host-bound loop it can be SLOWER than the real iterations)
            "fixed_ms_per_iter": fixed_ms, "fixed_share": min(1.0, fixed_ms / ms),
            "replicas_in_sync": in_sync,
            # the library's launches of one iteration (events around each; the torch glue between them is the rest)
            "kernels_ms": ktab, "kernels_ms_sum": None if ktab is None else round(sum(v["ms_per_iter"] for v in ktab.values()), 4),
            "roofline": roof,
            "config": {"workload": ("BASELINE configs[2]: legacy permutohash (24x2, 2^18) + MLP [128,128,64] appearance"
                                    if legacy else "SH neural-texture appearance (configs[1]'s model)")
                       + f", K={args.shells} subdiv-{args.subdiv} shells, {args.views} views of {args.res}x{args.res}"
                         f" in a TensorReel, dynamic ray count -> {args.target_hits} hits per iteration, "
                         "L1 + FusedAdam(0.9, 0.99, 1e-15) + warm-up 500, iterations "
                         f"{args.warmup}..{args.warmup + args.steps}",
                       "parameters": nparams, "parallelism": f"data-parallel x{world}"
                       + (", sharded Adam (reduce-scatter / slice update / all-gather)" if args.sharded_adam and world > 1 else "")},
        }
        out.update(dist_info(dist, args))
        print(json.dumps(out))


def run_dtu(args, world, rank, dev, dist):
    """BASELINE configs[3]'s per-GPU work: one 1600x1200 frame of a K=5 scene WITH the learned
    contracted background (`bg_color=None`: NerfHash field, 32 inverse-depth samples per ray through
    the packed ops; volsurfs.py:686-702, utils/background.py:31-141), forward + L1 + backward +
    fused Adam, in batches of 65 536 rays.  A step = one whole frame (30 batches)."""
    from volsurfs_amd.background import BoundingSphere
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    from volsurfs_amd.models import NerfHash
    from volsurfs_amd.trainer import train_step
    H, W, batch = 1200, 1600, 65536
    meshes = nested_shells(K=args.shells, subdiv=args.subdiv, device=dev)
    bg = NerfHash(3, "gridhash", "spherical_harmonics", device=dev)
    m = VolSurfs(meshes, max_rays=batch, bg_color=None, bg_model=bg,
                 bounding_primitive=BoundingSphere(0.5), nr_samples_bg=32, nr_warmup_iters=0)
    m.init_optim()
    o, d = pinhole_rays(H, W, focal=1111.1 * H / 800.0, cam_pos=(0.0, 0.0, -1.5), device=dev)
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    gt = torch.rand(H * W, 3, device=dev, generator=g)
    perm = torch.randperm(H * W, device=dev, generator=g)          # training draws random pixels
    o, d, gt = o[perm].contiguous(), d[perm].contiguous(), gt[perm].contiguous()
    N = H * W
    state = {"it": 0}

    def frame():
        for a in range(0, N, batch):
            m.grad_scale = GRAD_CHAIN_GAIN * float(min(batch, N - a))
            train_step(m, o[a:a + batch], d[a:a + batch], gt[a:a + batch], iter_nr=state["it"],
                       is_first_iter=state["it"] == 0, world=world, sync_losses=False)
            state["it"] += 1

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, min(args.warmup, 2))):
        frame()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ktab = roof = None
    if rank == 0 and dist is None:
        def one_batch():
            m.grad_scale = GRAD_CHAIN_GAIN * float(batch)
            train_step(m, o[:batch], d[:batch], gt[:batch], iter_nr=state["it"], is_first_iter=False, world=world,
                       sync_losses=False)
            state["it"] += 1
        ktab = kernel_table(one_batch, 3)
        # the fp32 matrix-core MLP of the background field: executed FLOPs of its two networks over the batch's samples
        dims = [[l.in_features, l.out_features] for mm in (bg.mlp_feat_and_density, bg.mlp_rgb)
                for l in mm.layers if isinstance(l, torch.nn.Linear)]
        fl_fwd = 2 * sum(a * b for a, b in dims) * batch * 32
        for key, mult in (("vsa_mlp_fwd", 1), ("vsa_mlp_bwd", 2)):
            if key in ktab:
                sec = ktab[key]["ms_per_iter"] * 1e-3
                ktab[key].update({"TFLOP/s": round(fl_fwd * mult / sec / 1e12, 2), "mfma_f32_frac": round(fl_fwd * mult / sec / 1e12 / 157.3, 4)})
        top = next(iter(ktab))
        roof = {"kernel": top, "kernel_ms": ktab[top]["ms_per_iter"], "peak_f32_mfma_TFLOP/s": 157.3,
                **{k: v for k, v in ktab[top].items() if k in ("TFLOP/s", "mfma_f32_frac")}}
    if rank == 0:
        nb = (N + batch - 1) // batch
        print(json.dumps({
            "kernels_ms_per_batch": ktab, "roofline": roof,
            "metric": "Mrays/s (fwd+bwd+Adam) at 1600x1200, K=5 shells + learned background",
            "value": N * world * args.steps / dt / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": min(args.warmup, 2), "ms_per_step": dt / args.steps * 1e3,
            "ms_per_batch": dt / args.steps / nb * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (background field) + f16 neural textures", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] per-GPU work: {W}x{H} frame, K={args.shells} subdiv-{args.subdiv} "
                                   "shells with SH neural textures, NerfHash background (3-D hash grid 24x2 2^18 + "
                                   "MLPs 51-64-64-64-65 / 80-64-64-3 on the fp32 matrix cores), 32 contracted samples "
                                   f"per ray through the packed ops, {nb} batches of {batch} random pixels per frame",
                       "bg_samples_per_frame": N * 32, "parallelism": f"data-parallel x{world}"},
            **dist_info(dist, args)}))


def run_render(args, world, rank, dev, dist):
    """Forward-only evaluation of one frame per step (base_method.py:366-541): the live model
    (trace -> unique texels -> encode -> MLP -> shade -> composite) and the deploy format
    (`bake()` once, then trace -> bilinear fetch of the 8-bit SH textures -> composite)."""
    from volsurfs_amd.camera import pinhole_rays
    from volsurfs_amd.mesh import nested_shells
    from volsurfs_amd.methods import VolSurfs
    res = args.res
    meshes = nested_shells(K=args.shells, subdiv=args.subdiv, device=dev)
    m = VolSurfs(meshes, max_rays=res * res)
    m.is_training = False
    o, d = pinhole_rays(res, res, focal=1111.1 * res / 800.0, cam_pos=(0.0, 0.0, -1.5), device=dev)

    def timed(fn):
        for _ in range(max(1, args.warmup)):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps
    t_live = timed(lambda: m.render(o, d, chunk=res * res))
    k_live = kernel_table(lambda: m.render(o, d, chunk=res * res), 3) if rank == 0 else None
    m.bake()
    t_baked = timed(lambda: m.render_baked(o, d))
    k_baked = kernel_table(lambda: m.render_baked(o, d), 3) if rank == 0 else None
    if rank == 0:
        n = res * res
        print(json.dumps({
            "kernels_ms": k_live, "kernels_ms_baked": k_baked,
            "metric": f"Mrays/s (forward only) at {res}x{res}, K={args.shells} shells", "value": n / t_live / 1e6,
            "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_live * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 neural textures / fp16 composite", "data": "synthetic",
            "baked": {"Mrays/s": n / t_baked / 1e6, "ms_per_frame": t_baked * 1e3,
                      "note": "8-bit baked SH textures (the deploy format): trace + shade + composite"},
            "config": {"workload": f"BaseMethod.render of one {res}x{res} frame, K={args.shells} subdiv-{args.subdiv} "
                                   "shells, SH neural textures, white background, one chunk"}}))


def orbit_views(o, d, n_views, deg):
    """n_views ray sets of the same pinhole camera orbiting the scene centre about the y axis,
    `deg` degrees apart (both origins and directions rotated)."""
    out = []
    for i in range(n_views):
        a = np.deg2rad(deg * i)
        R = torch.tensor([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]],
                         dtype=torch.float32, device=o.device)
        out.append(((o @ R.t()).contiguous(), (d @ R.t()).contiguous()))
    return out


def extra_scene(args, dev, use_graph, tag, steps=50, cold=False, orbit_deg=0.0, **synth):
    """The same step on another scene / under another regime, reported BESIDE the headline, never
    instead of it:

      noisy   radial noise 0.05, 6x6 randomly packed charts, U(+-1) table entries;
      stress  mesh.stress_shells: non-convex lobed shells (up to 6 crossings per ray), triangle areas
              spread 12x, 256 randomly packed charts per shell;
      cold    no inter-frame feedback at all — the traversal's stateless launch order
              (VSA_TRACE_FEEDBACK=0), the persistent kernels' fitted work split (VSA_NT_REBALANCE=0) —
              and a camera that orbits `orbit_deg` degrees per step, so that nothing a frame leaves
              behind fits the next one: what the FIRST frame of a new view costs.
    """
    from volsurfs_amd import neural_textures as _nt
    from volsurfs_amd.pipeline import KShellPipeline
    keep = _nt.REBALANCE
    if cold:
        _nt.REBALANCE = False
    try:
        pipe = KShellPipeline.synthetic(K=args.shells, subdiv=args.subdiv,
                                        res=args.res if args.width is None else (args.res, args.width), device=dev,
                                        seed=42, **synth)
    finally:
        _nt.REBALANCE = keep
    if cold:
        pipe.tracer.cost_feedback = False
    views = orbit_views(pipe.rays_o.clone(), pipe.rays_d.clone(), 8, orbit_deg) if orbit_deg else None
    for _ in range(3):
        pipe.step()
    run = pipe.step
    if use_graph:
        try:
            pipe.capture_graph()
            run = pipe.replay
        except Exception:
            run = pipe.step

    def frame(i):
        if views is not None:       # the next frame's rays arrive (15 MB device-to-device, inside the timed region)
            pipe.rays_o.copy_(views[i % len(views)][0])
            pipe.rays_d.copy_(views[i % len(views)][1])
        run()
    for i in range(24 if use_graph else 3):      # (untimed: the device clocks down during the capture, see the main line)
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        frame(3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pipe.stats()
    pipe.reset_stage_timers()
    pipe.step(record=True)
    st = pipe.stage_report()
    N = pipe.nr_rays
    rep = {"scene": pipe.scene_desc, "ms_per_step": dt * 1e3, "steps": steps,
           "h": (pipe.last_hits or 0) / float(N * pipe.K), "hits_per_frame": pipe.last_hits,
           "unique_texels_per_frame": pipe.last_slots, "Mhits/s": (pipe.last_hits or 0) / dt / 1e6,
           "trace_ms": round(st["trace"]["ms"], 4),
           "launch": "hip-graph replay" if run == getattr(pipe, "replay", None) and use_graph else "eager"}
    if cold:
        rep["regime"] = (f"VSA_TRACE_FEEDBACK=0, VSA_NT_REBALANCE=0, camera orbits {orbit_deg} deg per step "
                         "(8 views, rays copied in inside the timed region)")
    del pipe
    torch.cuda.empty_cache()
    return {"value_" + tag: N / dt / 1e6, tag: rep}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here —
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1, a free port — as a
    CHILD process (this parent has not touched the GPU and never does), pass rank 0's JSON line
    through and exit with the launcher's status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print(f"[bench] the {args.gpus}-rank launch failed (status {r.returncode}, {len(lines)} result lines)",
              file=sys.stderr)
        sys.exit(r.returncode or 1)
    print(lines[0])
    sys.exit(0)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            self_launch(args)          # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks",
              file=sys.stderr)
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank if (world > 1 and not args.single_device) else 0
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if world == 1:       # --force-dist: a one-rank group, rendezvous on a free local port
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(port))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        with stdout_to_stderr():
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), rank=rank, world_size=world)
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            warm = torch.zeros(1, device=torch.device("cuda", dev_index))
            dist.all_reduce(warm)              # the communicator (and its banner) comes up here at the latest
            torch.cuda.synchronize()
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    torch.manual_seed(42 + rank)
    if args.workload in ("train", "train-permuto", "dtu", "render"):
        {"dtu": run_dtu, "render": run_render}.get(args.workload, run_train)(args, world, rank, dev, dist)
        if dist is not None:
            dist.destroy_process_group()
        return

    from volsurfs_amd.pipeline import KShellPipeline
    strong = args.scaling == "strong" and world > 1
    rows = None
    if strong:       # every rank sees the same frame (same seed) and renders its own bands of it
        from volsurfs_amd.parallel import shard_bands
        rows = shard_bands(args.res, rank, world)
    if args.width is not None and strong:
        raise SystemExit("--width is for weak scaling / one GPU")
    from volsurfs_amd import neural_textures as _nt
    if args.cold:
        _nt.REBALANCE = False
    pipe = KShellPipeline.synthetic(K=args.shells, subdiv=args.subdiv,
                                    res=args.res if args.width is None else (args.res, args.width), device=dev,
                                    seed=42, gt_seed=42 if strong else 42 + rank, rows=rows,
                                    noise=args.noise, atlas_charts=args.atlas_charts, init=args.init,
                                    stress=args.stress)
    if args.cold:
        pipe.tracer.cost_feedback = False
    views = orbit_views(pipe.rays_o.clone(), pipe.rays_d.clone(), 8, args.orbit_deg) if args.cold and args.orbit_deg else None
    N = pipe.nr_rays

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from volsurfs_amd.parallel import GradientOverlap, OverlappedStep
    params = [pipe.bank.tables, pipe.bank.weights]
    wire = torch.bfloat16 if args.grad_wire_dtype == "bf16" else None
    overlap = GradientOverlap(world, wire_dtype=wire, force=args.force_dist)
    # the data-parallel step: ONE stream of launches — the same graph as on one GPU plus a
    # one-lane signal kernel — in which the device publishes "weights.grad final" and "phase p of
    # tables.grad final"; a side stream waits on those flags and all-reduces (sum) over RCCL / xGMI
    # while the rest of backward runs.  What is left exposed behind the last kernel is "grad_allreduce".
    if args.dp_reserve_cus is None:
        args.dp_reserve_cus = 16 if world > 1 else 0
    dp_on = (dist is not None or args.by_shell) and not args.dp_split_launches
    ostep = None
    if dp_on:
        phases = [int(x) for x in args.dp_phases.split(",")] if args.dp_phases else None
        ostep = OverlappedStep(pipe, world, wire_dtype=wire, force=args.force_dist, phases=phases,
                               wait_mode=args.dp_wait, reserve_cus=args.dp_reserve_cus, rank=rank,
                               direct_rccl={"auto": None, "rccl": True, "torch": False}[args.dp_comm]
                               if dist is not None else False)
        ostep.active = dist is not None
        dp_tuned = None
        if dist is not None and phases is None and args.dp_phases_auto:
            # the phase split from times measured on THIS group's wire (one phase when the reduction hides behind the
            # next step's head — a one-rank group then pays nothing for the schedule; parallel.choose_phases)
            _, dp_tuned = ostep.autotune_phases()

    def step(record=False):
        if ostep is not None:
            ostep.run(record=record)
            return
        if dist is None:
            pipe.step(record=record)
            return
        pipe.step(record=record, grad_ready=overlap.reduce_async)       # --dp-split-launches
        pipe.timer.run("grad_allreduce", overlap.wait, record,
                       bytes=sum(p.numel() for p in params) * 4)

    for _ in range(max(1, args.warmup)):
        step()
    pipe.stats()                       # hit / unique-texel counts for the byte & flop accounting
    # the stage / kernel tables below are taken at steady state: ten steps are 25 ms — neither the clocks nor the
    # measured-time work split of the persistent kernels (vsa_nt_rebalance) have settled by then (the same box
    # reported stage sums of 2.74 ms after 10 steps and 2.47 ms after 30, for a 2.44 ms step)
    for _ in range(max(0, 40 - args.warmup)):
        step()
    # per-kernel stage times: a few eager steps with events (outside the timed region)
    pipe.reset_stage_timers()
    for _ in range(3):
        step(record=True)
    stages = pipe.stage_report()
    # the dominant kernel's own duration (roofline.achieved): a second pass with events directly
    # around every C-ABI launch, each behind a short spin kernel so that the launch is already
    # queued when the start event is reached
    from volsurfs_amd import _lib
    _lib.kernel_events = {}
    for _ in range(3):
        step()
    kernel_ms = _lib.kernel_ms()
    _lib.kernel_events = None
    use_graph = not args.no_graph and not args.dp_split_launches
    if use_graph:
        try:
            # the step is ~25 launches on one stream with no host sync; the data-parallel step is the
            # same stream of launches (the collectives run beside the graph, released by device flags)
            if ostep is not None:
                # three graphs: the parameter-free head of a step (ray order, traversal, mark / compact) is
                # launched before the wait for the previous step's gradient reduction (OverlappedStep.run_split)
                pipe.capture_graph_split(dp=ostep.signals)
            else:
                pipe.capture_graph()
            # (the capture is ~0.1 s of host work with the GPU idle: untimed replays before the timed region, so that a
            #  short timed region — 20 steps = 50 ms — does not start on a device that has clocked down.  Same box, 20 timed
            #  steps: 2 replays 260.0 / 260.2 Mrays/s, 12: 263.3 / 265.3, 40: 267.1 = what 200 timed steps read)
            for _ in range(int(os.environ.get("VSA_BENCH_POST_CAPTURE_REPLAYS", "40"))):
                ostep.run_split(pipe.replay_prefix, pipe.replay_mid, pipe.replay_tail) if ostep is not None else pipe.replay()
            if ostep is not None:
                ostep.finish()
        except Exception as e:         # fall back to the (equally fast) eager path
            print(f"[bench] graph capture failed ({e}); timing the eager path", file=sys.stderr)
            use_graph = False

    def timed_step(i):
        if views is not None:
            pipe.rays_o.copy_(views[i % len(views)][0])
            pipe.rays_d.copy_(views[i % len(views)][1])
        if use_graph and ostep is not None:
            ostep.run_split(pipe.replay_prefix, pipe.replay_mid, pipe.replay_tail)
        elif use_graph:
            pipe.replay()
        else:
            step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed_step(i)
    if ostep is not None:
        ostep.finish()                 # the last step's reduction (inside the timed region)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # the reduced gradients of the last step, checksummed on EVERY rank: after the all-reduce they are the same bits
    # everywhere, so the first multi-GPU run validates its own collectives (outside the timed region)
    grad_cs = None
    if dist is not None:
        b = pipe.bank
        cs = torch.stack([b.tables.grad.double().sum(), b.tables.grad.double().abs().sum(),
                          b.weights.grad.double().sum(), b.weights.grad.double().abs().sum()])
        got = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(got, cs)
        rows = [[float(v) for v in g.cpu()] for g in got]
        grad_cs = {"grad_checksum_per_rank": rows, "grad_checksums_equal": all(r == rows[0] for r in rows),
                   "grad_checksum_is": "[sum, sum |.|] of tables.grad and of weights.grad after the last step's all-reduce"}

    if rank == 0:
        total_rays = (pipe.loss_rays if strong else N * world) * args.steps
        value = total_rays / dt / 1e6
        # dominant KERNEL (every stage of the step is one library launch since round 4), by the kernels' own
        # durations (events directly around the launches): a stage's eager time also holds the host's gaps
        dom = max(((k, v) for k, v in stages.items() if k != "grad_allreduce" and "vsa_" + k in kernel_ms),
                  key=lambda kv: kernel_ms["vsa_" + kv[0]])
        name, st = dom
        # PMC-measured HBM bytes per launch (tools/traffic.sh -> profiles/traffic.json), valid only
        # for the kernel sources they were collected at: a stale file is reported as null
        traffic, traffic_note, traffic_all = None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            tj = json.load(open(tf)).get("workloads", {}).get(workload_key(args))
            if tj is None:
                traffic_note = "profiles/traffic.json holds no collection on this workload (tools/traffic.sh <tag> <bench args>)"
            elif tj.get("_kernel_source_sha256") != kernel_source_hash():
                traffic_note = "profiles/traffic.json was collected at other kernel sources (stale): re-run tools/traffic.sh"
            else:
                traffic = tj.get(name)
                traffic_all = {k: v for k, v in tj.items() if not k.startswith("_")}
        # the kernel's own duration: events directly around its launch (a stage also holds the
        # small torch kernels next to it); one launch per stage except the encode stages
        k_ms = kernel_ms.get("vsa_" + name, st["ms"])
        if st.get("flops") and st.get("bound") == "mfma":
            # executed, unpadded FLOPs of the texels this launch evaluates
            ach = st["flops"] / (k_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": name, "kernel_ms": k_ms, "achieved": ach,
                    "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F16_PEAK_TFLOPS,
                    "traffic": traffic, "hbm_GB/s": st["bytes"] / (k_ms * 1e-3) / 1e9,
                    # `frac` counts the executed UNPADDED FLOPs of the texels (a third of this kernel's matrix work is
                    # the forward recompute, a fifth of the rest W3 padded to 32 rows: even a saturated pipe reads 0.53);
                    # `pmc` = what the matrix pipe itself saw (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs))
                    "pmc": pmc_busy(args, name)}
        else:
            ach = st["bytes"] / (k_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "kernel_ms": k_ms, "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic}
        out = {
            "metric": "Mrays/s (fwd+bwd) at 800x800, K=5 shells" if (args.res, args.width, args.shells) == (800, None, 5)
            else f"Mrays/s (fwd+bwd) at {args.width or args.res}x{args.res}, K={args.shells} shells",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": pipe.dtype_desc, "data": "synthetic",
            "config": dict(pipe.config_desc(world), launch="hip-graph replay" if use_graph else "eager",
                           **({"dp_phases": ostep.signals.phase_end,
                               "dp_phases_chosen_from": dp_tuned if dp_tuned else "given / the fixed default",
                               "dp_reserve_cus": ostep.signals.reserve_cus,
                               "dp_comm": "rccl called directly on the side stream" if ostep.rccl is not None
                               else "torch.distributed",
                               "dp_schedule": "three graphs per step: the parameter-free head (ray order, traversal, "
                                              "mark/compact) runs before the wait for the previous step's gradient "
                                              "reduction; then zero_grad .. MLP backward; then the hash-grid backward, "
                                              "beside which a side stream all-reduces each phase as its device flag rises"}
                              if ostep is not None else {}),
                           scene=getattr(pipe, "scene_desc", None)),
            "roofline": roof,
            # hit fraction per (ray, shell) and the rate per hit: 71 % of the (ray, shell) pairs of this
            # frame are misses that cost almost nothing, so Mrays/s alone overstates the shading rate
            "h": (pipe.last_hits or 0) / float(N * pipe.K),
            "Mhits/s": (pipe.last_hits or 0) * world * args.steps / dt / 1e6 if not strong else None,
            "stages_ms": {k: round(v["ms"], 4) for k, v in stages.items()},
            "stage_roofline": {k: pipe.stage_roofline(v, HBM_PEAK_GBS, MFMA_F16_PEAK_TFLOPS)
                               for k, v in stages.items()},
        }
        if traffic_note:
            out["roofline"]["traffic_note"] = traffic_note
        if traffic_all:
            out["traffic_bytes_per_launch"] = traffic_all          # PMC, every kernel of the step on this workload
        if dist is None and pipe.tracer.node_format == "q16":
            out["stage_roofline"]["trace"].update(trace_ceiling(pipe, kernel_ms.get("vsa_trace_q_fb", stages["trace"]["ms"]), args))
        if world == 1 and dist is None and not args.no_noisy and not (args.noise or args.atlas_charts or args.stress or args.cold):
            out.update(extra_scene(args, dev, use_graph, "noisy", noise=0.05, atlas_charts=6, init="spread"))
            out.update(extra_scene(args, dev, use_graph, "stress", stress=True, init="spread"))
            out.update(extra_scene(args, dev, use_graph, "cold", cold=True, orbit_deg=args.orbit_deg))
            out.update(extra_scene(args, dev, use_graph, "stress_cold", stress=True, init="spread", cold=True,
                                   orbit_deg=args.orbit_deg))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_sample"] = cpu_baseline(pipe, args.cpu_sample_rays)
        out.update(dist_info(dist, args))
        if grad_cs is not None:
            out.update(grad_cs)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
