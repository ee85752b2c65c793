#!/usr/bin/env python
"""bench.py — render+backward frames/s of the MI355X rasterizer on BASELINE.json's metric config.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script re-executes itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
(one rank per GPU, RCCL); launched by torchrun directly it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

A FRAME is `render(camera, gaussians, bg)` through the reference's Python API (activations inside the HIP preprocess
kernels by default, `--torch-activations` for the reference's stock PyTorch ops) followed by the backward pass down to the
raw Gaussian parameters, replayed as one HIP graph.
  N = 1: a step is `--rounds` (4) rounds of the frames in flight: the metric's configuration, configs[1].  HOW the
         frames are in flight is calibrated (40 steps each; `config.in_flight_calibration_frames_per_s`): 1, 2 or 3 views
         with a stream, a handle and a graph each, or 2 / 3 batched LAUNCH CHAINS of 4 views each (fr_forward_batch /
         fr_backward_batch: every kernel of the frame launched once for the chain's views; a stream and a graph per
         chain) — `--chains off` keeps the per-view streams, `--chains 4x3` forces a shape.  Every view has its own
         camera, handle, scratch buffers and gradient buffers either way; `one_frame_at_a_time`, `views_on_streams` and
         `batched_views` in the line are the other modes' rates on the same box.
  N > 1: TWO modes are timed in the same run, each with its own K steps, and both are in the line (`dp.modes`):
    literal    BASELINE configs[3] as written — every rank renders ONE view of the replicated Gaussians, then ONE RCCL
               all-reduce(AVG) of the flat gradient buffer (in place, on the view's stream), and only then the next step
               starts (a synchronous data-parallel optimisation step: the next frame needs the averaged gradient's
               update).  THIS is `value`.
    amortised  `--rounds` x (views in flight) frames per rank and exchange, their mean folded into one of two exchange
               buffers and all-reduced while the next step's frames render (local gradient accumulation; step k+2 waits
               for the collective of step k; the timed region ends when the last one has finished).
  The un-overlapped all-reduce time is reported for both payloads of SURVEY.md §8e: the generic parameter set (59 floats
  per Gaussian, 23.6 MB at 100 k) and FateAvatar's 'gs' group (12 floats, 4.8 MB; train/optim.py:15-21).
Workload (config.workload): BASELINE.json configs[1] — 100 000 Gaussians sampled on the head template, 512x512,
SH degree 3 (M=16), synthetic data, random-init appearance.  Inputs are resident in HBM before the timed region.

The JSON line carries
  roofline     — the blend backward (the graded kernel): algorithmic bytes 76*R + 20*H*W + 8*T (SURVEY.md §8d;
                 R = num_rendered, T = 16x16 tiles, reference semantics) / that kernel's mean launch duration, measured
                 with HIP events on the launch stream (handed to the dispatch itself, hipExtLaunchKernelGGL: the
                 kernel's execution as a profiler times it) over eager launches of the same frame right after the timed
                 region (events inside a replayed graph cannot be read back), against the 8 TB/s HBM peak.  `traffic`
                 and `valu_frac` are NOT measured by this run: they come from the committed counter profile named in
                 `counters_source` (rocprofv3 --pmc passes, tools/pmc.sh), or are null when no profile matches.
  stage_frac   — every stage's algorithmic bytes / its measured time / 8 TB/s (formulas: STAGE_BYTES below).
  cpu_baseline — the CPU oracle (oracle/fr_oracle.c, OpenMP, kind "port": the reference has no CPU rasterizer) on a
                 bounded sample of the same frames: best thread count of a quick sweep, and one thread.
  dp           — N > 1: ranks seen, backend, RCCL version, per-rank num_rendered, both modes' frame rates, and the
                 un-overlapped all-reduce duration per payload.
  dp_reference_at_1 — N = 1: the two N > 1 steps measured on a one-rank RCCL group (this script again, as a child
                 process with --exchange-at-1, after the timed region): `value` at N = 1 is frames in flight with no
                 exchange at all, `value` at N > 1 the literal configs[3] step — a scaling curve over `value` compares the
                 two; the like-for-like first point of the N > 1 series is `dp_reference_at_1.literal.value`.

Diagnostics: FR_BENCH_HOST_TIME=1 prints how long the host took to enqueue a step; FR_BENCH_ORDER=coherent stores the scene's
Gaussians in a spatially coherent order (not the metric's scene: what a coherent input order would be worth); FR_BENCH_BATCH_X_STREAMS=1 adds more
(views per chain, chains) shapes to `batched_views`.
FR_BENCH_STUB=1 replaces the rasterizer by a deterministic CPU gradient generator so that the N > 1 control flow can
be exercised without GPUs (tests/test_bench_dp.py); such a line says "data": "stub" and is not a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The ROCm runtime switches of fateavatar_amd.tune_runtime() (fateavatar_amd/__init__.py: kernel arguments of eager launches
# in device memory; replayed graphs enqueue their nodes like ordinary launches) — set HERE, explicitly, before torch
# initialises the runtime, and reported in config.hip_env.  FR_BENCH_RUNTIME_DEFAULTS=1 leaves the runtime alone: the
# default run measures one frame at a time that way too, as a child process (`runtime_defaults` in the line).
if os.environ.get("FR_BENCH_RUNTIME_DEFAULTS") != "1":
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s achievable)
STUB = os.environ.get("FR_BENCH_STUB") == "1"


def stage_bytes(P: int, M: int, R: int, H: int, W: int) -> dict:
    """Algorithmic (compulsory) bytes of each stage; binning + sort counted with ONE read + write of
    the 12-byte key/value pairs (the reference's radix sort makes ~6)."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {"preprocess_fwd": P * (44 + 12 * M) + P * 75,
            "binning_sort": P * 24 + R * 12 * 3 + R * 8 + T * 8,   # stages scan + emit + tile_sort together
            "blend_fwd": 40 * R + 20 * H * W + 8 * T,
            "blend_bwd": 76 * R + 20 * H * W + 8 * T,
            "preprocess_bwd": P * (111 + 12 * M) + P * (64 + 12 * M)}


def stage_bytes_required(P: int, M: int, R: int, H: int, W: int) -> dict:
    """Where this implementation's data layout needs fewer bytes than SURVEY.md's formula bills: the per-Gaussian backward
    reads the forward's 36-byte d colour / d direction instead of its 12 M-byte SH row: P * 147 in,
    P * (64 + 12 M) out.  The other stages' formulas are what they must move."""
    sb = stage_bytes(P, M, R, H, W)
    sb["preprocess_bwd"] = P * 147 + P * (64 + 12 * M)
    return sb


def cpu_baseline(scene, budget_s: float = 20.0):
    """Oracle forward+backward on the host: a quick sweep over thread counts picks the best; one thread beside it."""
    from oracle import oracle
    c = scene.camera
    H, W = c.image_height, c.image_width
    dpix = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    kw = dict(bg=scene.bg, means3D=scene.means3D, opacities=scene.opacities, viewmatrix=c.world_view_transform,
              projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx, tanfovy=c.tanfovy, H=H,
              W=W, shs=scene.shs, sh_degree=scene.sh_degree, scales=scene.scales, rotations=scene.rotations)

    def run(n):
        t0 = time.perf_counter()
        for _ in range(n):
            oracle.backward(oracle.forward(**kw), dpix)
        return time.perf_counter() - t0

    ncpu = oracle.num_threads()
    quota = oracle.cpu_quota_cores()   # what the container may use (16 CPUs on the GPU boxes, of 256 hardware threads)
    run(1)  # page-in, thread pool
    sweep = {}
    for t in sorted({quota, max(1, quota // 2), min(ncpu, 2 * quota), min(ncpu, 4 * quota)}, reverse=True):
        oracle.set_num_threads(t)
        run(1)
        sweep[t] = run(2) / 2
    best = min(sweep, key=sweep.get)
    oracle.set_num_threads(best)
    n = int(max(4, min(64, (budget_s * 0.5) / sweep[best])))
    dt = run(n)
    oracle.set_num_threads(1)
    n1 = int(max(1, min(8, (budget_s * 0.3) / max(run(1), 1e-3))))
    dt1 = run(n1)
    oracle.set_num_threads(ncpu)
    # `cores` = the CPUs the container may use (its quota) that the best run kept busy; `threads` = the OpenMP threads of that run
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": min(best, quota), "threads": best, "kind": "port",
            "sample": f"{n} frames fwd+bwd of the same workload at {best} threads ({dt:.1f} s); thread sweep "
                      + ", ".join(f"{k}: {1 / v:.2f}" for k, v in sorted(sweep.items())) + " frames/s",
            "value_1thread": round(n1 / dt1, 3), "host_threads": ncpu, "cpu_quota_cores": quota,
            "note": "the container's CPU quota bounds the port: thread counts beyond `cpu_quota_cores` time-slice against each other"}


# ---------------------------------------------------------------- engines: what a "frame" is
class _View:
    """One view of the step: its camera, its replica of the Gaussians (parameters and gradient buffer), its upstream
    gradient, and — views are IN FLIGHT TOGETHER — its own stream, fr_handle slot and captured graph."""

    def __init__(self, eng, k, scene):
        from fateavatar_amd.model import FlatGaussians, TorchCamera
        if os.environ.get("FR_BENCH_ORDER") == "coherent":   # (diagnostic: the same Gaussians stored in a spatially coherent
            from fateavatar_amd.scenes import spatial_order           # order (scenes.spatial_order) instead of the metric's random one)
            o = spatial_order(scene.means3D)
            for a in ("means3D", "scales", "rotations", "opacities", "shs"):
                setattr(scene, a, np.ascontiguousarray(getattr(scene, a)[o]))
        self.k, self.scene = k, scene
        # N > 1: two replicas (and two captured graphs), used by alternate steps, so that the gradient exchange of step i
        # reads one set of gradient buffers while the frames of step i + 1 already write the other
        self.pcs = [FlatGaussians(scene.means3D, scene.shs, scene.opacities, scene.scales, scene.rotations, scene.sh_degree,
                                  eng.dev, fused_activations=eng.args.fused_activations)
                    for _ in range(2 if eng.exchanging else 1)]
        self.pc = self.pcs[0]
        self.cam = TorchCamera(scene.camera, eng.dev)
        self.bg = torch.from_numpy(scene.bg).to(eng.dev)
        H = W = eng.args.res
        # upstream gradient of the image: d(L1-mean against a fixed random target)/d(pixel) has magnitude 1/(3HW) and
        # a random sign; it is fixed, so the step is exactly render + backward
        g = ((torch.rand((3, H, W), generator=torch.Generator().manual_seed(1 + eng.rank * 64 + k)) < 0.5).float() * 2 - 1)
        self.dL_dpix = (g / (3 * H * W)).to(eng.dev)
        self.stream = None            # (HipEngine picks streams that really overlap: fateavatar_amd/streams.py)
        self.done = torch.cuda.Event()
        self.graphs = [None] * len(self.pcs)       # the frame that OVERWRITES the set's gradient buffers
        self.graphs_add = [None] * len(self.pcs)   # (N > 1, rounds > 1) the frame that ADDS to them
        self.graph = None


class HipEngine:
    """The product path: FlatGaussians + render() + autograd backward on one MI355X, each view replayed as a HIP graph.
    `--in-flight K`: K views of the step run concurrently (a stream, an fr_handle and a graph each).  One frame's
    kernels are latency-bound at this size — 1.5 waves per SIMD in the per-Gaussian kernels, 4 workgroups in the totals
    kernel — and leave most of the chip idle; the reference's step loops over its batch of views one after the other
    (model/fateavatar.py:251-276), here the views of the batch overlap."""

    def __init__(self, args, rank, world, local):
        from fateavatar_amd import rasterizer, scenes
        from fateavatar_amd.render import render
        self.rasterizer, self.local, self.args, self.rank = rasterizer, local, args, rank
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.exchanging = world > 1 or args.exchange_at_1   # the views' completion events and second buffer sets are only needed by the exchange
        self.rounds = max(1, args.rounds)
        self.round_no = 0
        self.auto = args.in_flight <= 0          # pick the number of views in flight by a short calibration run
        K = self.K = 3 if self.auto else args.in_flight
        # replicated Gaussians (same seed everywhere), view index (rank * K + k) of world * K views around the head
        self.views = [_View(self, k, scenes.head_scene(P=args.P, res=args.res, sh_degree=args.sh_degree, seed=0,
                                                       view=rank * K + k, n_views=max(world * K, 1), scale=args.scale,
                                                       opacity=args.opacity)) for k in range(K)]
        self.scene = self.views[0].scene
        from fateavatar_amd.streams import concurrent_streams
        for v, st in zip(self.views, concurrent_streams(K, self.dev, also_with=[torch.cuda.current_stream(self.dev)])):
            v.stream = st
        self.all_views = self.views
        self._render = render
        self.graph = None
        self.grads_read = [None, None]   # events: the exchange has read the gradient buffers of set 0 / 1
        self._chains, self._extra_views = {}, []
        self.chains = None               # (K, streams, views, graphs) when the timed steps run batched launch chains

    def set_mode(self, K, rounds):
        """Which views take part in a step and how many rounds a step has (bench modes: literal = 1 view, 1 round)."""
        self.views = self.all_views[:K]
        self.K, self.rounds, self.round_no = K, max(1, rounds), 0
        self.grads_read = [None, None]

    def frame(self, v=None, which=0, add=False):
        v = v or self.views[0]
        pc = v.pcs[which]
        if not add:
            pc.begin_step()                        # grads set to None: backward assigns (zero_grad(set_to_none=True))
        else:
            pc.accumulate_into_kept_grads()
        # (add: the gradients of the previous round are kept, and the backward kernel adds this frame's to them in
        # place — FR_FLAG_ACCUMULATE through the parameters' gradient slots: local gradient accumulation)
        out = self._render(v.cam, pc, v.bg)                        # activations + HIP rasterizer forward
        torch.autograd.backward(out["render"], grad_tensors=v.dL_dpix)  # HIP rasterizer backward (+ activations)

    def prepare(self):
        """Per view: eager warm-up (sizes the binning capacity, fills the allocator pools), then capture ONE frame."""
        for v in self.views:
            with self.rasterizer.handle_slot(v.k):
                for _ in range(max(3, self.args.warmup // 2)):
                    self.frame(v)
                torch.cuda.synchronize()
                if self.args.graph:
                    # the per-frame work is launch-bound on the host (~40 small launches): capture one frame and replay
                    # it; the rasterizer runs in no-wait mode inside the capture (no host synchronisation at all);
                    # overflow of the binning capacity is checked after the timed region
                    with self.rasterizer.no_wait():
                        for which in range(len(v.pcs)):
                            with torch.cuda.stream(v.stream):
                                for _ in range(3):
                                    self.frame(v, which)
                            torch.cuda.synchronize()
                            v.graphs[which] = torch.cuda.CUDAGraph()
                            # thread_local: the RCCL watchdog thread of an N > 1 run must not trip the capture
                            with torch.cuda.graph(v.graphs[which], stream=v.stream, capture_error_mode="thread_local"):
                                self.frame(v, which)
                            if self.exchanging and self.rounds > 1:
                                with torch.cuda.stream(v.stream):
                                    self.frame(v, which, add=True)
                                torch.cuda.synchronize()
                                v.graphs_add[which] = torch.cuda.CUDAGraph()
                                with torch.cuda.graph(v.graphs_add[which], stream=v.stream, capture_error_mode="thread_local"):
                                    self.frame(v, which, add=True)
                    torch.cuda.synchronize()
            v.graph = v.graphs[0]
        self.graph = self.views[0].graph

    def batch_frame(self, views):
        """The views' frames through ONE launch chain (render_batch -> fr_forward_batch / fr_backward_batch)."""
        from fateavatar_amd.render import render_batch
        for v in views:
            v.pc.begin_step()
        outs = render_batch([v.cam for v in views], [v.pc for v in views], [v.bg for v in views], slots=[v.k for v in views])
        torch.autograd.backward([o["render"] for o in outs], grad_tensors=[v.dL_dpix for v in views])

    def setup_chains(self, K, streams):
        """`streams` launch chains of K views each (fr_forward_batch / fr_backward_batch: every kernel of the frame launched
        once for the K views), every chain captured as one graph on a stream of its own.  Returns (views, [(graph, stream)])."""
        from fateavatar_amd import scenes
        from fateavatar_amd.streams import concurrent_streams
        key = (K, streams)
        if key in self._chains:
            return self._chains[key]
        n = K * streams
        views = list(self.all_views[:n])
        views += self._extra_views[:n - len(views)]
        while len(views) < n:   # (more views than the stream mode uses: further cameras around the head)
            k = len(self.all_views) + len(self._extra_views)
            v = _View(self, k, scenes.head_scene(P=self.args.P, res=self.args.res, sh_degree=self.args.sh_degree, seed=0,
                                                 view=self.rank * n + k, n_views=max(n, 1), scale=self.args.scale, opacity=self.args.opacity))
            self._extra_views.append(v)
            views.append(v)
        groups = [views[i * K:(i + 1) * K] for i in range(streams)]
        sts = concurrent_streams(streams, self.dev, also_with=[torch.cuda.current_stream(self.dev)]) if streams > 1 \
            else [torch.cuda.Stream(device=self.dev)]
        graphs = []
        for grp, side in zip(groups, sts):
            for _ in range(3):
                self.batch_frame(grp)
            torch.cuda.synchronize()
            with self.rasterizer.no_wait():
                side.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self.batch_frame(grp)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    self.batch_frame(grp)
            graphs.append((g, side))
        torch.cuda.synchronize()
        self._chains[key] = (views, graphs)
        return self._chains[key]

    @staticmethod
    def replay_chains(graphs):
        for g, side in graphs:
            with torch.cuda.stream(side):
                g.replay()

    def chains_overflowed(self, views):
        for v in views:
            with self.rasterizer.handle_slot(v.k):
                if self.rasterizer.check_async_overflow(self.local):
                    return True
        return False

    def measure_batched(self, K, reps, streams=1):
        """K views per launch chain and captured graph, `streams` such chains in flight on streams of their own:
        frames/s (None if it cannot be set up here)."""
        views, graphs = self.setup_chains(K, streams)
        for _ in range(10):
            self.replay_chains(graphs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.replay_chains(graphs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.chains_overflowed(views):
            return None
        return {"value": round(K * streams * reps / dt, 2), "unit": "frames/s", "views_per_launch_chain": K, "streams": streams,
                "launches_per_frame": round(6.0 / K, 2), "us_per_launch_chain": round(dt / reps * 1e6, 1)}

    def calibrate(self, world):
        """--in-flight 0 (default): how many views should be in flight?  It depends on how the runtime maps streams to
        hardware queues (three is best with its default four queues), so it is measured: 40 steps with 1,
        2 and 3 views, rates summed over the ranks, best count kept — the same on every rank."""
        self.calibration = None
        if not self.auto:
            return
        rates = []
        for k in range(1, len(self.views) + 1):
            for _ in range(5):
                self.enqueue_frame(self.views[:k], count=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                self.enqueue_frame(self.views[:k], count=False)
            torch.cuda.synchronize()
            rates.append(40 * k / (time.perf_counter() - t0))
        r = torch.tensor(rates, dtype=torch.float64, device=self.dev)
        if world > 1:
            torch.distributed.all_reduce(r)
        self.K = int(torch.argmax(r).item()) + 1
        self.views = self.views[:self.K]
        self.all_views = self.views
        self.calibration = {str(k + 1): round(float(x) / world, 1) for k, x in enumerate(r.tolist())}
        # ... or batched launch chains in flight together: K views per chain through the same six launches
        # (fr_forward_batch / fr_backward_batch), `streams` chains on streams of their own.  One process only: the
        # exchange modes keep a graph per view (their gradient sets alternate per step).
        if world == 1 and not self.exchanging and self.args.graph and self.args.chains != "off":
            best = float(r.max().item())
            cands = [(4, 2), (4, 3)] if self.args.chains == "auto" else [tuple(int(x) for x in self.args.chains.split("x"))]
            for K, S in cands:
                try:
                    m = self.measure_batched(K, 30, S)
                except Exception as e:   # (a chain that cannot be set up here must not cost the run: the per-view streams remain)
                    print(f"[bench] {S} chains x {K} views could not be set up ({type(e).__name__}: {e}); skipped", file=sys.stderr)
                    m = None
                if m is None:
                    continue
                self.calibration[f"{S} chains x {K} views"] = m["value"]
                if m["value"] > best * 1.02 or self.args.chains != "auto":
                    best = m["value"]
                    views, graphs = self.setup_chains(K, S)
                    self.chains = (K, S, views, graphs)

    def enqueue_frame(self, views=None, count=True):
        """One ROUND: every view's render + backward, each on its own stream.  N > 1: a step is `rounds` rounds on one
        set of gradient buffers (the first overwrites them, the others add), and steps alternate between the two sets."""
        if self.chains is not None and views is None and count:
            self.replay_chains(self.chains[3])   # one ROUND = every chain's K views once
            self.round_no += 1
            return
        which, add, last = 0, False, False
        if self.exchanging and count:
            step, r = divmod(self.round_no, self.rounds)
            which, add, last = step & 1, r > 0, r == self.rounds - 1
        self.which = which
        for v in (views or self.views):
            with torch.cuda.stream(v.stream):
                if not add and self.grads_read[which] is not None:
                    v.stream.wait_event(self.grads_read[which])   # (N > 1) the exchange of two steps ago read this set
                g = v.graphs_add[which] if add else v.graphs[which]
                if g is not None:
                    g.replay()
                else:
                    with self.rasterizer.handle_slot(v.k):
                        self.frame(v, which, add)
                if last:                                   # the exchange waits for the step's last round only
                    v.done.record(v.stream)
        if count:
            self.round_no += 1

    def flat_grads(self):
        """Gradient buffers of the step enqueued last."""
        return [v.pcs[getattr(self, "which", 0)].collect_grads() for v in self.views]

    def mark_grads_read(self):
        """(current stream) everything enqueued so far has read the gradient buffers of the step enqueued last."""
        ev = torch.cuda.Event()
        ev.record()
        self.grads_read[getattr(self, "which", 0)] = ev

    def join(self):
        """Make the current stream wait for the step's views (the exchange reads their gradients)."""
        cur = torch.cuda.current_stream(self.dev)
        for v in self.views:
            cur.wait_event(v.done)

    def sync(self):
        torch.cuda.synchronize()

    def run_avatar_mode(self, args, world):
        """The third N > 1 mode: `--steps` FateAvatar optimisation steps (AvatarStep: the exchange captured in the step's
        graph on the nccl backend), one frame per rank and step."""
        import importlib.util
        from fateavatar_amd import dp
        spec = importlib.util.spec_from_file_location("fr_train_synthetic", os.path.join(ROOT, "tools", "train_synthetic.py"))
        ts = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ts)
        try:
            su = ts.fateavatar_setup(args.P, args.res, self.dev, use_graph=args.graph)
            st, cams, posed, gts, nf = su["st"], su["cams"], su["posed"], su["gts"], su["n_frames"]

            def step(it):
                f = (it * world + self.rank) % nf
                st.step(cams[f], posed[f], gts[f])

            for it in range(max(args.warmup, 10)):
                step(it)
            torch.cuda.synchronize()
            dp.barrier()
            t0 = time.perf_counter()
            for it in range(args.steps):
                step(args.warmup + it)
            torch.cuda.synchronize()
            dp.barrier()
            dt = time.perf_counter() - t0
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            if world > 1:
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
            st.check()
            return {"frames_per_step_per_gpu": 1, "rounds_per_step": 1, "frames_in_flight_per_gpu": 1, "overlap": False,
                    "ms_per_step": round(dt / args.steps * 1e3, 4), "value": round(world * args.steps / dt, 2), "unit": "frames/s",
                    "steps_per_s": round(args.steps / dt, 2), "allreduce_payload_bytes": 12 * args.P * 4,
                    "exchange_in_graph": bool(getattr(st, "exchange_in_graph", False)), "sh_degree": 0,
                    "binding": "inside the per-Gaussian kernels (fr_aux::binding)",
                    "storage_order": "the reference's UV-raster initialisation (uniform_sampling_barycoords on the template's UV "
                                     "layout, model/fateavatar.py:128-133; tools/train_synthetic.py --random-order is the A/B)",
                    "step": "bind + render + L1 + backward + all-reduce(AVG) of the 'gs' group + densification statistics + Adam"}
        except Exception as e:   # (the headline modes must not be lost to this one)
            return {"status": "failed: " + repr(e)[:300]}

    def finish(self):
        """After the timed region: overflow check of the captured frames, per-kernel durations, counts."""
        from fateavatar_amd import _lib
        if self.chains is not None and self.chains_overflowed(self.chains[2]):
            raise SystemExit("binning capacity overflowed inside a captured launch chain; rerun (capacity hint was raised)")
        for v in self.views:
            with self.rasterizer.handle_slot(v.k):
                if v.graph is not None and self.rasterizer.check_async_overflow(self.local):
                    c = self.rasterizer.last_counts[self.local]
                    raise SystemExit(f"binning capacity overflowed inside the captured graph (instances {c.num_instances}, "
                                     f"num_rendered {c.num_rendered}, max list {c.max_tile_list}); rerun (capacity hint was raised)")
        # HIP events around every stage launch, on the launch stream, over eager frames of ONE view (event records inside
        # a replayed graph cannot be read back)
        # (in no-wait mode, like the captured frames: the host does not block on each frame's counts, so the launches
        # queue up behind each other as they do in a replayed graph instead of each starting on an idle, cold GPU)
        _lib.profile_enable(self.local, True)
        with self.rasterizer.no_wait():
            for _ in range(min(self.args.steps, 50)):
                self.frame()
        torch.cuda.synchronize()
        prof = _lib.profile_read(self.local)
        _lib.profile_enable(self.local, False)
        self.frame()                             # (one waiting frame: its counts)
        torch.cuda.synchronize()
        c = self.rasterizer.last_counts[self.local]
        return prof, dict(num_rendered=int(c.num_rendered), num_instances=int(c.num_instances), max_tile_list=int(c.max_tile_list))


class StubEngine:
    """FR_BENCH_STUB=1: a deterministic gradient per (rank, step) on the CPU instead of the rasterizer, so that the
    launch / exchange / timing control flow of this file runs without a GPU.  Not a measurement."""

    def run_avatar_mode(self, args, world):
        from fateavatar_amd import dp
        buf = torch.zeros(12 * 64, dtype=torch.float32)
        dp.barrier()
        t0 = time.perf_counter()
        for it in range(args.steps):
            buf.fill_(float(self.rank + it))
            dp.allreduce_mean_async(buf).wait()
        dp.barrier()
        dt = time.perf_counter() - t0
        return {"frames_per_step_per_gpu": 1, "rounds_per_step": 1, "frames_in_flight_per_gpu": 1, "overlap": False,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "value": round(world * args.steps / dt, 2), "unit": "frames/s",
                "steps_per_s": round(args.steps / dt, 2), "allreduce_payload_bytes": buf.numel() * 4, "exchange_in_graph": False,
                "sh_degree": 0, "step": "stub", "mean_check": float(buf[0])}

    def __init__(self, args, rank, world, local):
        self.dev = torch.device("cpu")
        self.rank, self.k = rank, 0
        self.rounds = max(1, args.rounds)
        self.K = max(1, args.in_flight)
        self.calibration = None
        self.grads = self.all_grads = [torch.zeros(1 << 12) for _ in range(self.K)]
        self.scene = None
        self.grads_read = None

    def prepare(self):
        pass

    def calibrate(self, world):
        pass

    def set_mode(self, K, rounds):
        self.K, self.rounds, self.k = K, max(1, rounds), 0
        self.grads = self.all_grads[:K]

    def enqueue_frame(self, views=None, count=True):
        for j, g in enumerate(self.grads):   # like the engine: a step's first round overwrites, the others add
            x = torch.arange(g.numel(), dtype=torch.float32) * 1e-3 + (self.rank * self.K + j + 1) * (self.k + 1)
            g.copy_(x) if self.k % self.rounds == 0 else g.add_(x)
        self.k += 1

    def flat_grads(self):
        return self.grads

    def join(self):
        pass

    def mark_grads_read(self):
        pass

    def sync(self):
        pass

    def finish(self):
        return {}, dict(num_rendered=1000 + self.rank, num_instances=0, max_tile_list=0)


class GradExchange:
    """All-reduce(AVG) of the flat gradient of every step, overlapped with the following frames: the gradients of the
    step's views are summed (behind their backward passes) into one of two exchange buffers and reduced there
    asynchronously; a buffer is reused two steps later, after waiting for its collective."""

    def __init__(self, like: torch.Tensor, rounds: int = 1):
        self.bufs = [torch.empty_like(like), torch.empty_like(like)]
        self.works = [None, None]
        self.k = 0
        self.rounds = max(1, int(rounds))   # every view's buffer holds the sum over the step's rounds
        self.last = self.bufs[0]

    def submit(self, eng, serial=False):
        """After the step's last round: mean of the views' accumulated gradients into an exchange buffer, all-reduce.
        `serial` (the literal step: one view, one round, nothing overlapped): the view's gradient buffer is reduced in
        place, on the view's own stream, and that stream waits for the collective before its next frame — frame,
        all-reduce, frame, all-reduce ..., two stream hand-overs per step (to RCCL's stream and back) and no copy."""
        from fateavatar_amd import dp
        grads = eng.flat_grads()
        if serial and len(grads) == 1 and self.rounds == 1:
            st = getattr(eng.views[0], "stream", None) if hasattr(eng, "views") else None
            if st is not None:
                with torch.cuda.stream(st):
                    dp.allreduce_mean_async(grads[0]).wait()
            else:                                    # (the CPU stub of tests/test_bench_dp.py)
                dp.allreduce_mean_async(grads[0]).wait()
            self.last = grads[0]
            self.k += 1
            return
        i = self.k & 1
        if self.works[i] is not None:
            self.works[i].wait()
        eng.join()                                   # the step's views have written their gradients
        grads = eng.flat_grads()
        buf = self.bufs[i]
        scale = 1.0 / (len(grads) * self.rounds)     # mean over the rank's frames of the step; the collective averages over the ranks
        if len(grads) == 1 and scale == 1.0:
            buf.copy_(grads[0], non_blocking=True)
        elif buf.is_cuda:                            # one pass
            from fateavatar_amd.loss import scaled_sum
            scaled_sum(buf, grads, scale)
        else:                                        # (the CPU stub of tests/test_bench_dp.py)
            acc = grads[0].clone()
            for g in grads[1:]:
                acc.add_(g)
            torch.mul(acc, scale, out=buf)
        eng.mark_grads_read()                        # the step after next may overwrite these gradient buffers
        self.works[i] = dp.allreduce_mean_async(buf)
        self.last = buf
        self.k += 1

    def drain(self):
        for i in (0, 1):
            if self.works[i] is not None:
                self.works[i].wait()
                self.works[i] = None

    def latest(self) -> torch.Tensor:
        return self.last


def _flush_c_stdio():
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child_line(extra, timeout=240):
    """This script again as a child process of ONE rank (its own HIP context), after the timed region: its JSON line."""
    # (not torchrun's environment: the child is a world of its own — with TORCHELASTIC_USE_AGENT_STORE set it would wait for
    # the parent job's rendezvous store)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
           and not k.startswith(("TORCHELASTIC_", "ROLE_", "GROUP_", "TORCH_NCCL_", "NCCL_ASYNC"))}
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"child bench exited {out.returncode}: {(out.stderr or out.stdout)[-300:]}")
    return json.loads(lines[-1])


def _scene_args(args):
    a = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--P", str(args.P), "--res", str(args.res),
         "--sh-degree", str(args.sh_degree), "--rounds", str(args.rounds)]
    if args.scale is not None:
        a += ["--scale", str(args.scale)]
    return a


def _dp_reference_at_1(args):
    """`bench.py --exchange-at-1` as a child process (its own HIP context and one-rank RCCL group), after this process's timed
    region: the N > 1 steps at one rank.  A run that cannot produce it says why in `status`."""
    try:
        rec = _child_line(["--exchange-at-1", "--cpu-seconds", "0", "--no-opaque", "--no-coherent", "--no-runtime-defaults", "--no-config5", "--opacity", str(args.opacity)] + _scene_args(args))
        m = rec["dp"]["modes"]
        keys = ("value", "unit", "ms_per_step", "frames_per_step_per_gpu")
        return {"status": "ok",
                "what": "the N > 1 steps on a one-rank RCCL group (`bench.py --exchange-at-1`, a second process after the timed region)",
                "literal": {k: m["literal"][k] for k in keys}, "amortised": {k: m["amortised"][k] for k in keys},
                "avatar": None if not m.get("avatar") else {k: m["avatar"].get(k) for k in keys + ("steps_per_s", "allreduce_payload_bytes")},
                "allreduce_table": rec["dp"]["allreduce_table"], "rccl_version": rec["dp"]["rccl_version"]}
    except Exception as e:   # (no RCCL, no free port, ...)
        return {"status": "failed: " + repr(e)[:300]}


def _coherent_layout(args):
    """What a CALLER gets by storing the same Gaussians in a spatially coherent order (scenes.spatial_order; the reference's
    UV-raster initialisation produces one): this run again as a child process with FR_BENCH_ORDER=coherent.  Not the metric's
    scene (its order is random) and never `value`."""
    try:
        os.environ["FR_BENCH_ORDER"] = "coherent"
        rec = _child_line(["--cpu-seconds", "0", "--no-dp-reference", "--no-opaque", "--no-coherent", "--no-runtime-defaults", "--no-config5", "--opacity", str(args.opacity)] + _scene_args(args))
        return {"status": "ok", "value": rec["value"], "unit": rec["unit"], "one_frame_at_a_time": (rec.get("one_frame_at_a_time") or {}).get("value"),
                "preprocess_fwd_us": (rec.get("stage_us") or {}).get("preprocess_fwd"),
                "what": "the same Gaussians stored in grid-cell order (fateavatar_amd.scenes.spatial_order), same modes as `value`"}
    except Exception as e:
        return {"status": "failed: " + repr(e)[:300]}
    finally:
        os.environ.pop("FR_BENCH_ORDER", None)


def _runtime_defaults(args):
    """One frame at a time with the ROCm runtime's OWN defaults (neither switch of fateavatar_amd.tune_runtime() set): this
    script again as a child process with FR_BENCH_RUNTIME_DEFAULTS=1."""
    saved = {k: os.environ.pop(k, None) for k in ("HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
    try:
        os.environ["FR_BENCH_RUNTIME_DEFAULTS"] = "1"
        rec = _child_line(["--in-flight", "1", "--cpu-seconds", "0", "--no-dp-reference", "--no-opaque", "--no-coherent", "--no-runtime-defaults", "--no-config5",
                           "--opacity", str(args.opacity)] + _scene_args(args)[:-2] + ["--rounds", "1"])
        return {"status": "ok", "one_frame_at_a_time": rec["value"], "unit": rec["unit"], "hip_env": rec["config"]["hip_env"],
                "what": "the same frames, one at a time, in a process that sets neither runtime switch"}
    except Exception as e:
        return {"status": "failed: " + repr(e)[:300]}
    finally:
        os.environ.pop("FR_BENCH_RUNTIME_DEFAULTS", None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def _opaque_scene(args):
    """The operating point training moves to (opacity 0.9; config/fateavatar.yaml:40-47 prunes below 0.005, the rest
    saturates): the same scene and run, one frame at a time, as a child process — frames/s and the blend backward's launch."""
    try:
        rec = _child_line(["--opacity", "0.9", "--in-flight", "1", "--cpu-seconds", "0", "--no-dp-reference", "--no-opaque", "--no-coherent", "--no-runtime-defaults", "--no-config5"]
                          + _scene_args(args)[:-2] + ["--rounds", "1"])
        r = rec.get("roofline") or {}
        return {"status": "ok", "opacity": 0.9, "value": rec["value"], "unit": rec["unit"], "frames_in_flight": 1,
                "num_rendered": rec["config"]["num_rendered"], "blend_bwd_us": r.get("avg_launch_us"), "blend_bwd_frac": r.get("frac"),
                "stage_us": rec.get("stage_us")}
    except Exception as e:
        return {"status": "failed: " + repr(e)[:300]}


def _config5_scene(args):
    """SURVEY.md §8d config 5 — 500 000 Gaussians at 1024 x 1024, the largest single-GPU configuration, where the launch ramp
    distorts the graded kernel's fraction least — as a child process after the timed region: the blend backward's isolated
    launch (dispatch-tied events, as `roofline`), its algorithmic bytes (76 R + 20 HW + 8 T) and fraction, every stage's
    duration and fraction, one frame at a time."""
    try:
        rec = _child_line(["--P", "500000", "--res", "1024", "--in-flight", "1", "--cpu-seconds", "0", "--no-dp-reference", "--no-opaque",
                           "--no-coherent", "--no-runtime-defaults", "--no-config5", "--steps", str(min(args.steps, 60)),
                           "--warmup", str(min(args.warmup, 10)), "--sh-degree", str(args.sh_degree), "--rounds", "1"], timeout=600)
        r = rec.get("roofline") or {}
        return {"status": "ok", "workload": rec["config"]["workload"], "bound": "hbm", "kernel": r.get("kernel"),
                "avg_launch_us": r.get("avg_launch_us"), "launches": r.get("launches"), "algorithmic_bytes": r.get("algorithmic_bytes"),
                "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"),
                "num_rendered": rec["config"]["num_rendered"], "tile_instances_8x8": rec["config"]["tile_instances_8x8"],
                "max_tile_list": rec["config"]["max_tile_list"],
                "one_frame_at_a_time": rec["value"], "stage_us": rec.get("stage_us"), "stage_frac": rec.get("stage_frac")}
    except Exception as e:
        return {"status": "failed: " + repr(e)[:300]}


def stock_run(args):
    return args.scale is None and args.opacity == 0.1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--P", type=int, default=100_000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--scale", type=float, default=None, help="splat scale (default: the template's NN spacing at P)")
    ap.add_argument("--opacity", type=float, default=0.1, help="splat opacity (the metric's scene: 0.1)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="views of a step that run concurrently on one GPU: 1 = one frame at a time (as in round 1), 0 = "
                         "calibrate (1, 2 or 3, whichever renders most frames per second here)")
    ap.add_argument("--chains", default="auto",
                    help="N = 1: batched launch chains in flight ('KxS' = S chains of K views each, 'auto' = try 4x2 and 4x3 "
                         "next to the per-view streams and keep the fastest, 'off' = per-view streams only)")
    ap.add_argument("--rounds", type=int, default=4,
                    help="rounds of views per step: a step renders rounds x (views in flight) frames per GPU and, on N > 1 "
                         "GPUs, exchanges their mean gradient ONCE (local gradient accumulation in the backward kernel, "
                         "the usual way to keep a 23.6 MB all-reduce behind the rendering).  The same at every N")
    ap.add_argument("--exchange-at-1", action="store_true",
                    help="diagnostic: run N = 1 with the N > 1 machinery (second gradient-buffer sets, per-round fold, "
                         "all-reduce on a one-rank RCCL group): what the exchange costs the rendering apart from the wire")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-dp-reference", dest="dp_reference", action="store_false", default=True,
                    help="N = 1: do not add `dp_reference_at_1` (the N > 1 step — one view, one all-reduce on a one-rank RCCL "
                         "group — measured by a second run of this script after the timed region)")
    ap.add_argument("--no-opaque", dest="opaque", action="store_false", default=True,
                    help="N = 1: do not add `opaque` (the same scene at opacity 0.9, one frame at a time, measured by a second run "
                         "of this script after the timed region)")
    ap.add_argument("--no-coherent", dest="coherent", action="store_false", default=True,
                    help="N = 1: do not add `coherent_layout` (the same Gaussians stored in a spatially coherent order, measured by a "
                         "second run of this script after the timed region)")
    ap.add_argument("--no-runtime-defaults", dest="runtime_defaults", action="store_false", default=True,
                    help="N = 1: do not add `runtime_defaults` (one frame at a time in a process that leaves the ROCm runtime's "
                         "switches alone, measured by a second run of this script after the timed region)")
    ap.add_argument("--no-config5", dest="config5", action="store_false", default=True,
                    help="N = 1: do not add `roofline_config5` (the blend backward's launch and every stage at 500 k Gaussians / "
                         "1024 x 1024, measured by a second run of this script after the timed region)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True, help="replay the frame as a HIP graph")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", default=True,
                    help="N > 1: wait for every all-reduce before the next frame starts")
    ap.add_argument("--fused-activations", dest="fused_activations", action="store_true", default=True,
                    help="sigmoid/exp/normalize inside the HIP kernels (SURVEY.md §8f row 1)")
    ap.add_argument("--torch-activations", dest="fused_activations", action="store_false",
                    help="reference behaviour: activations as stock PyTorch ops")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher (one rank per GPU of this node)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        raise SystemExit(subprocess.call(cmd, env=env))

    from fateavatar_amd import dp
    if args.exchange_at_1:
        os.environ["FR_DP_GROUP_OF_ONE"] = "1"
    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not STUB and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")

    eng = (StubEngine if STUB else HipEngine)(args, rank, world, local)
    eng.prepare()
    eng.calibrate(world)
    exchanging = world > 1 or args.exchange_at_1
    K_best = eng.K

    def run_mode(K, rounds, overlap):
        """`--warmup` untimed and `--steps` timed steps of one mode; returns (seconds, max over the ranks; exchange)."""
        eng.set_mode(K, rounds)
        xchg = GradExchange(eng.flat_grads()[0], rounds) if exchanging else None

        def step():
            for _ in range(rounds):
                eng.enqueue_frame()
            if xchg is not None:
                xchg.submit(eng, serial=not overlap)
                if not overlap:
                    xchg.drain()

        for _ in range(args.warmup):
            step()
        if xchg is not None:
            xchg.drain()
        eng.sync()
        dp.barrier()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if os.environ.get("FR_BENCH_HOST_TIME") and rank == 0:
            print(f"[host] K={K} rounds={rounds}: the {args.steps} steps were enqueued in {(time.perf_counter() - t0) * 1e6 / args.steps:.1f} us each", file=sys.stderr)
        if xchg is not None:
            xchg.drain()          # the last collectives are inside the clock
        eng.sync()
        dp.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=eng.dev)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item()), xchg

    rounds = max(1, args.rounds)
    modes = {}
    if exchanging:
        # BASELINE configs[3] as written: one view per rank, exchange, then the next step (see the module docstring)
        el, xl = run_mode(1, 1, False)
        modes["literal"] = {"frames_per_step_per_gpu": 1, "rounds_per_step": 1, "frames_in_flight_per_gpu": 1, "overlap": False,
                            "ms_per_step": round(el / args.steps * 1e3, 4), "value": round(world * args.steps / el, 2),
                            "unit": "frames/s", "grad_checksum": float(xl.latest().double().abs().sum().item())}
        ea, xchg = run_mode(K_best, rounds, args.overlap)
        modes["amortised"] = {"frames_per_step_per_gpu": K_best * rounds, "rounds_per_step": rounds,
                              "frames_in_flight_per_gpu": K_best, "overlap": bool(args.overlap),
                              "ms_per_step": round(ea / args.steps * 1e3, 4),
                              "value": round(world * K_best * rounds * args.steps / ea, 2), "unit": "frames/s"}
        elapsed, frames_per_step = el, 1
        # ... and the reference's OWN step (train/optim.py:15-21, train/iteration.py:50-60): FateAvatar's mesh-bound
        # parameters at SH degree 0 — the 'gs' group, 12 floats per Gaussian (4.8 MB at 100 k) — one frame per rank:
        # bind, render, L1, backward, all-reduce(AVG), Adam as ONE captured graph, the collective a node of it
        modes["avatar"] = eng.run_avatar_mode(args, world)
    else:
        elapsed, xchg = run_mode(K_best, rounds, False)
        frames_per_step = K_best * rounds
        if getattr(eng, "chains", None) is not None:
            frames_per_step = eng.chains[0] * eng.chains[1] * rounds

    # the same engine with ONE view in flight (what `value` measured until views overlapped): a reference point
    single = None
    if not STUB and K_best > 1 and not exchanging:
        eng.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps * rounds):
            eng.enqueue_frame(eng.views[:1])
        eng.sync()
        single = {"value": round(args.steps * rounds / (time.perf_counter() - t1), 2), "unit": "frames/s", "frames_in_flight": 1}

    # K views through ONE launch chain (render_batch): the in-flight gain without any stream / hardware-queue arrangement
    batched = None
    if not STUB and not exchanging and args.graph:
        combos = [(2, 1), (3, 1), (4, 1)]
        if os.environ.get("FR_BENCH_BATCH_X_STREAMS"):   # (experiment: batched chains in flight together)
            combos += [(4, 2), (3, 3), (4, 3), (4, 4), (2, 3)]
        batched = [b for b in (eng.measure_batched(K, max(20, args.steps * rounds // 2), S) for K, S in combos) if b is not None]

    prof, counts = eng.finish()

    # ---- data-parallel record: who took part, what was exchanged, how long one exchange takes on its own
    dpinfo = None
    if exchanging:
        import torch.distributed as dist
        reduced = xchg.latest().clone()

        def time_allreduce(numel):
            buf = torch.zeros(numel, dtype=reduced.dtype, device=reduced.device)
            for _ in range(3):
                dp.allreduce_mean_async(buf).wait()
            eng.sync()
            dp.barrier()
            t1 = time.perf_counter()
            n_ar = 20
            for _ in range(n_ar):
                dp.allreduce_mean_async(buf).wait()
            eng.sync()
            sec = (time.perf_counter() - t1) / n_ar
            nbytes = numel * buf.element_size()
            return {"payload_bytes": nbytes, "us": round(sec * 1e6, 1),
                    "busbw_GBps": round(2 * (world - 1) / world * nbytes / sec / 1e9, 1)}

        # the payload the steps exchanged (the generic set: 59 floats per Gaussian at SH degree 3), and FateAvatar's own
        # optimizer group (opacity 1 + offset 1 + colour 3 + rotation 4 + scaling 3 = 12 floats, train/optim.py:15-21)
        table = [dict(time_allreduce(reduced.numel()), what="flat gradient of the timed steps (generic parameter set)")]
        if not STUB:
            table.append(dict(time_allreduce(12 * args.P), what="FateAvatar 'gs' optimizer group, 12 floats per Gaussian"))
        mine = torch.tensor([counts["num_rendered"]], dtype=torch.int64, device=eng.dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rccl = None
        if dist.get_backend() == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        dpinfo = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                  "num_rendered_per_rank": [int(t.item()) for t in allr],
                  "allreduce_payload_bytes": table[0]["payload_bytes"], "allreduce_us": table[0]["us"],
                  "allreduce_busbw_GBps": table[0]["busbw_GBps"], "allreduce_table": table,
                  "group_of_one": bool(args.exchange_at_1),
                  "overlap": bool(args.overlap),
                  "modes": modes,
                  "value_is": "modes.literal (BASELINE configs[3]: one view per rank per exchange, no overlap with the next step)",
                  "grad_checksum": float(reduced.double().abs().sum().item())}
        # The generic literal step is wire-bound by construction: every gradient array leaves one kernel at the end of the
        # step and the next forward needs the update, so nothing can overlap the exchange.  Its ceiling, from this run's own
        # numbers: frame / (frame + un-overlapped all-reduce of the same payload).
        frame_us = modes["literal"]["ms_per_step"] * 1e3 - table[0]["us"]
        dpinfo["literal_ceiling"] = {"frame_us": round(max(frame_us, 0.0), 1), "allreduce_us": table[0]["us"],
                                     "efficiency_ceiling": round(max(frame_us, 0.0) / max(modes["literal"]["ms_per_step"] * 1e3, 1e-9), 4),
                                     "what": "frame_us / (frame_us + allreduce_us): the scaling efficiency the literal mode can reach "
                                             "against the same step without a wire (frame_us = its step time minus the un-overlapped "
                                             "all-reduce of its payload, allreduce_table[0])"}

    if rank == 0:
        fps = world * frames_per_step * args.steps / elapsed
        H = W = args.res
        M = (args.sh_degree + 1) ** 2
        R = counts["num_rendered"]
        roof = stage_frac = stage_frac_required = stages = None
        if prof:
            sb = stage_bytes(args.P, M, R, H, W)
            stages = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}
            us = dict(stages)
            us["binning_sort"] = sum(us.get(k, 0.0) for k in ("scan", "emit", "tile_sort"))
            stage_frac = {k: round(sb[k] / (us[k] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) for k in sb if us.get(k)}
            sbr = stage_bytes_required(args.P, M, R, H, W)
            stage_frac_required = {k: round(sbr[k] / (us[k] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) for k in sbr if us.get(k)}
            ms, n = prof["blend_bwd"]
            if n:
                avg_s = ms / n * 1e-3
                ach = sb["blend_bwd"] / avg_s / 1e9
                traffic = valu = valu4 = valu_meas = src = None
                cpath = os.path.join(ROOT, "profiles", "blend_bwd_counters.json")
                if os.path.exists(cpath):   # written by tools/pmc.sh + tools/pmcstats.py from rocprofv3 --pmc passes
                    try:
                        cj = json.load(open(cpath))
                        if cj.get("P") == args.P and cj.get("res") == args.res and args.scale is None and args.opacity == 0.1:
                            traffic, src = cj.get("hbm_bytes_per_launch"), "profiles/blend_bwd_counters.json (" + str(cj.get("collected")) + ")"
                            if cj.get("sq_insts_valu_per_launch"):
                                # MI355X_MICROARCH.md: a wave64 VALU op issues over 2 cycles on CDNA4's SIMD-32 ->
                                # wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel time)
                                valu = round(cj["sq_insts_valu_per_launch"] * 2 / (1024 * 2.4e9 * avg_s), 4)
                                # (the conservative reading — 4 cycles per wave64 op, SIMD-16 style — beside it: the in-repo
                                # microbenchmark tools/diag/micro_issue.hip measures 2.5 - 2.9 cycles per wave-instruction
                                # for this mix, i.e. between the two)
                                valu4 = round(cj["sq_insts_valu_per_launch"] * 4 / (1024 * 2.4e9 * avg_s), 4)
                                # ... and at the issue rate this kernel's instruction mix was MEASURED to sustain once 2-3
                                # waves share a SIMD: 1.3 ns per wave-instruction (tools/diag/micro_issue.hip)
                                valu_meas = round(cj["sq_insts_valu_per_launch"] * 1.3e-9 / (1024 * avg_s), 4)
                    except Exception:
                        pass
                roof = {"bound": "hbm", "kernel": "k_unit_blend_bwd_sparse", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": traffic, "valu_frac": valu,
                        "valu_frac_at_4_cycles_per_op": valu4, "valu_frac_measured_issue": valu_meas,
                        "counters_source": src, "algorithmic_bytes": sb["blend_bwd"],
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches": n,
                        "measured": "dispatch-tied HIP events around ISOLATED launches (eager no-wait frames of one view, queued "
                                    "behind each other, nothing else on the GPU); rocprofv3 of `bench.py --in-flight 1` agrees (profiles/)"}
        cpu = None
        if args.cpu_seconds > 0 and world == 1 and eng.scene is not None:
            cpu = cpu_baseline(eng.scene, args.cpu_seconds)
        # What `value` is on N > 1 GPUs (the literal configs[3] step) measured at ONE rank, so that a scaling curve has a
        # like-for-like first point: `value` at N = 1 is frames in flight without any exchange, a different mode.
        dp_ref = opaque = None
        if not exchanging or world > 1:
            if args.dp_reference and args.graph and (not STUB or world > 1):
                dp_ref = _dp_reference_at_1(args)
        if not STUB and world == 1 and not exchanging and args.opaque and args.graph and args.opacity == 0.1:
            opaque = _opaque_scene(args)
        rt_defaults = None
        if not STUB and world == 1 and not exchanging and args.runtime_defaults and args.graph and stock_run(args):
            rt_defaults = _runtime_defaults(args)
        config5 = None
        if not STUB and world == 1 and not exchanging and args.config5 and args.graph and stock_run(args) and (args.P, args.res) == (100_000, 512):
            config5 = _config5_scene(args)
        coherent = None
        if (not STUB and world == 1 and not exchanging and args.coherent and args.graph and args.scale is None
                and os.environ.get("FR_BENCH_ORDER") is None):
            coherent = _coherent_layout(args)
        # the like-for-like first point of a scaling curve over the N > 1 `value` (the literal step), and this run against it
        scaling_reference = efficiency = None
        if dp_ref and dp_ref.get("status") == "ok":
            scaling_reference = dp_ref["literal"]["value"]
            if world > 1:
                efficiency = round(fps / (world * scaling_reference), 4)
                for name in ("amortised", "avatar"):
                    if modes.get(name) and modes[name].get("value") and dp_ref.get(name) and dp_ref[name].get("value"):
                        modes[name]["efficiency"] = round(modes[name]["value"] / (world * dp_ref[name]["value"]), 4)
        stock = args.scale is None and args.opacity == 0.1
        cfg_name = ("BASELINE.json configs[1]" if (args.P, args.res) == (100_000, 512) and stock else
                    "SURVEY.md §8d config 5 (not the metric's configuration)" if (args.P, args.res) == (500_000, 1024) and stock
                    else f"custom scene (scale {args.scale}, opacity {args.opacity}; not the metric's configuration)")
        chains = getattr(eng, "chains", None)
        line = {
            "metric": "render+backward frames/sec at 512^2, 100k Gaussians", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "stub" if STUB else "synthetic",
            "config": {"workload": f"{cfg_name}: {args.P} Gaussians on the head template, "
                                   f"{args.res}x{args.res}, SH deg {args.sh_degree} (M={M}), "
                                   "forward+backward through render() with a fixed dL/dpixel",
                       "frames_per_step_per_gpu": frames_per_step,
                       "rounds_per_step": 1 if exchanging else rounds,
                       "frames_in_flight_per_gpu": 1 if exchanging else (chains[0] * chains[1] if chains else K_best),
                       "launch_chains_in_flight": None if not chains else {"chains": chains[1], "views_per_chain": chains[0]},
                       "in_flight_calibration_frames_per_s": eng.calibration,
                       "activations": "fused in the HIP preprocess kernels" if args.fused_activations else "stock PyTorch",
                       "launch": "hipgraph replay" if args.graph else "eager",
                       "hip_env": {k: os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")},
                       "parallelism": (f"dp{world}: one view per GPU and step, then ONE flat-grad RCCL all-reduce(AVG), then the next "
                                       "step (BASELINE configs[3]); the amortised mode is in dp.modes" if exchanging else
                                       (f"dp1 ({rounds} round(s) per step of {chains[1]} launch chains in flight, {chains[0]} views per chain "
                                        "(fr_forward_batch / fr_backward_batch), a stream and a graph per chain)" if chains else
                                        f"dp1 ({rounds} round(s) of {K_best} view(s) in flight per step, each view on its own stream)")),
                       "num_rendered": R, "tile_instances_8x8": counts["num_instances"],
                       "max_tile_list": counts["max_tile_list"]},
            "roofline": roof,
            # the same kernel at SURVEY.md §8d config 5 (500 k Gaussians, 1024 x 1024: the largest single-GPU configuration)
            "roofline_config5": config5,
            "cpu_baseline": cpu, "stage_us": stages, "stage_frac": stage_frac,
            # (stage_frac bills SURVEY.md's formulas; stage_frac_required the bytes this layout has to move: stage_bytes_required)
            "stage_frac_required": stage_frac_required, "dp": dpinfo,
            "one_frame_at_a_time": single,
            # ... and in a process that sets neither ROCm runtime switch (config.hip_env is what THIS process ran under)
            "runtime_defaults": rt_defaults,
            # the N > 1 `value` (literal step) at ONE rank: divide an N > 1 run's `value` by N times this, not by the N = 1 `value`
            "scaling_reference": scaling_reference, "scaling_reference_status": None if dp_ref is None else dp_ref.get("status"),
            "efficiency": efficiency,
            "dp_reference_at_1": dp_ref,
            # the scene at opacity 0.9 (where training takes the Gaussians), one frame at a time
            "opaque": opaque,
            # the same Gaussians in a spatially coherent storage order (what a caller's layout is worth; never `value`)
            "coherent_layout": coherent,
            # (when `value` is the launch-chain mode) the best per-view-stream count of the calibration: a stream, a handle
            # and a graph per view, as `value` was measured until round 3
            "views_on_streams": (None if not chains or not eng.calibration else
                                 {"value": max(v for k, v in eng.calibration.items() if k.isdigit()), "unit": "frames/s",
                                  "views_in_flight": K_best, "measured": "40-step calibration run"}),
            # the same frames with K views per launch chain on ONE stream (fr_forward_batch / fr_backward_batch); `value`
            # above is the stream mode (a stream, a handle and a graph per view)
            "batched_views": batched,
            # every stage's algorithmic bytes over the measured frame rate: the whole path against HBM
            "frame_roofline": (None if not prof else {
                "algorithmic_bytes_per_frame": int(sum(sb.values())), "achieved": round(sum(sb.values()) * fps / world / 1e9, 1),
                "unit": "GB/s per GPU", "frac": round(sum(sb.values()) * fps / world / 1e9 / HBM_PEAK_GBPS, 4)}),
        }
    else:
        line = None
    # RCCL prints a version banner through C stdio, which (into a pipe) is flushed only at exit — behind the JSON line.
    # Everything buffered goes out first, on every rank, so that the JSON line is the last line of the run.
    _flush_c_stdio()
    if torch.distributed.is_initialized():
        dp.barrier()
        torch.distributed.destroy_process_group()
        _flush_c_stdio()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
