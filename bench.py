#!/usr/bin/env python3
"""bench.py -- IRR-PWC train-step throughput on MI355X (BASELINE.json metric: image-pairs/sec fwd+bwd).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either started by ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or from a plain shell: the script then starts those N ranks itself as
child processes (before anything touches the GPU in the parent), relays rank 0's JSON line and exits with their return code.

A "step" is one pass of the hot path over one synthetic batch: zero_grad -> PWCNet forward (train mode) ->
MultiScaleEPE_PWC_Bi_Occ_upsample -> NaN check -> backward -> [RCCL gradient all-reduce] -> Adam step,
i.e. the reference's TrainingEpoch._step (runtime.py:131-194) with inputs already resident in HBM.
Workload = BASELINE configs[2] (384x448, 32 pairs per GPU, FlyingChairsOcc-shaped synthetic tensors,
weak scaling: every rank gets its own 32 pairs).  Prints ONE JSON line on rank 0.  After the headline timing the same process
runs 5 steps of the per-GPU workload of BASELINE configs[4] (448x1024, 8 pairs per GPU: north_star's second crop) and reports it
under ``"secondary"`` (own roofline object); ``metric`` / ``value`` are the headline's.

Further keys of the line (rank 0, N = 1): ``reference_harness`` -- the same model under the REFERENCE's own loop, literally
(runtime.py:158-189: ``optimizer.zero_grad()``, forward, ``.item()`` NaN assertion before ``backward()``, stock
``torch.optim.Adam.step()``; no GradArena / FusedAdam / TrainStep), as a user who swaps the model class into runtime.py gets it
(the model installs arena + lane itself, irr_amd/harness.py) and with that switched off (``plain_autograd``); ``forward_only`` --
BASELINE configs[1] (eval forward, 8 pairs of 384x448) with its EPE against the oracle on the first two pairs;
``roofline.single_stream`` -- the dominant kernel over 3 steps without the second stream, same process (kernel quality without
lane time-sharing).  ``--harness reference`` makes the reference loop the HEADLINE instead (A/B runs).

``IRR_DDP_BACKEND=gloo`` (single-GPU boxes, tests): the ranks share the visible GPUs (rank r -> cuda:r % device_count) and
exchange through gloo -- RCCL refuses two ranks on one device; everything but the transport is the same code.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide: v_mfma_f32_32x32x16_bf16, dense
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0   # conv_x3: six bf16 MFMA products per fp32 product (exact 3-way split)
H2_PEAK_TFLOPS = 2500.0 / 3.0          # the fp16x2 form of conv_x3: three fp16 MFMA products per fp32 product (dense fp16 = dense bf16 peak)
X3_PLANES = {1: 352, 2: 616, 3: 640}


def kernel_name(var):
    """variant code of irr_amd.conv's KernelTimer -> (the kernel's template instantiation SPELLED AS rocprofv3 PRINTS IT, so that the
    line's ``roofline.kernel`` is a row of profiles/*_kernel_stats.txt; its MFMA roof in fp32 TFLOP/s; "hbm" | "mfma" = what bounds it)"""
    if var >= 100000 and 9010 <= var % 100000 <= 9014:          # the streaming 32-channel kernel, one instantiation per epilogue form
        np_ = 2 if var >= 200000 else 3
        return (f"conv_x3s_kernel<{var % 10}, {np_}>", H2_PEAK_TFLOPS if np_ == 2 else X3_PEAK_TFLOPS, "hbm")
    if var >= 200000:
        c = var - 200000
        return (f"conv_x3_kernel<{c // 1000}, {(c // 100) % 10}, {(c // 10) % 10}, {X3_PLANES.get(c % 10, 0)}, 2>", H2_PEAK_TFLOPS, "mfma")
    if var >= 100000:
        c = var - 100000
        return (f"conv_x3_kernel<{c // 1000}, {(c // 100) % 10}, {(c // 10) % 10}, {X3_PLANES.get(c % 10, 0)}, 3>", X3_PEAK_TFLOPS, "mfma")
    return (f"conv_fwd_kernel<{var // 100}, {(var // 10) % 10}, {var % 10}>", FP32_MFMA_PEAK_TFLOPS, "mfma")


HBM_PEAK_TBPS = 8.0                    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E
CONV_GFLOP_PER_PAIR = {(384, 448): 1088.4, (448, 1024): 2902.5}   # 3x forward conv FLOPs
SECONDARY = (8, 448, 1024)             # per-GPU share of BASELINE configs[4] (Sintel-shaped 448x1024, bs64 on 8 GPUs)
SECONDARY_STEPS = 5
TRAFFIC_FILES = {(384, 448, 32): "hbm_traffic.json", (448, 1024, 8): "hbm_traffic_448x1024.json"}


NBATCHES = 4                           # distinct synthetic batches resident in HBM, fed round-robin (no step sees the batch of the step before)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def physical_cores():
    """physical cores of the host (sockets x cores per socket from /proc/cpuinfo; half the logical CPUs as a fallback)"""
    try:
        seen = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def synthetic_batch(batch, height, width, seed, device):
    """inputs U[0,1), targets 5*N(0,1) px, occlusion Bernoulli(0.2); CPU generator, then copied."""
    g = torch.Generator().manual_seed(seed)
    b = {"input1": torch.rand(batch, 3, height, width, generator=g),
         "input2": torch.rand(batch, 3, height, width, generator=g),
         "target1": 5 * torch.randn(batch, 2, height, width, generator=g),
         "target2": 5 * torch.randn(batch, 2, height, width, generator=g),
         "target_occ1": (torch.rand(batch, 1, height, width, generator=g) < 0.2).float(),
         "target_occ2": (torch.rand(batch, 1, height, width, generator=g) < 0.2).float()}
    return {k: v.to(device) for k, v in b.items()}


def _cpu_leg(O, height, width, batch_pairs, threads, warm=2, timed=5):
    """median step time of the oracle's train step at one (batch, thread count) setting -> image-pairs/s"""
    torch.set_num_threads(threads)
    P = O.make_trainable(O.synthetic_params(0))
    opt = O.make_adam(P)
    batch = O.synthetic_batch(batch_pairs, height, width, 1234)
    for _ in range(warm):
        O.train_step(P, opt, batch)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        O.train_step(P, opt, batch)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return batch_pairs / ts[len(ts) // 2]


def cpu_baseline(height, width, quick=False):
    """The oracle's train step (CPU restatement of the reference path, kind="port") timed on the host cores of the GPU box
    with the protocol of BASELINE.md section 3: >= 2 warm-up + >= 5 timed steps, median step time; batch 2 and batch 8; the
    fastest thread count for this graph on the host (16: probed 8/16/32/64/128 threads -> 0.52/0.59/0.46/0.22/0.07 pairs/s
    on the 128-core EPYC 9575F box, tools/cpu_threads_probe.py -- more threads are SLOWER) and the 8-thread figure that is
    comparable with the survey container.  ``value`` = the best of the legs.  (~2.5 min of CPU work; ``quick``: one leg.)"""
    from oracle import irr_pwc_oracle as O
    ncpu = os.cpu_count() or 16
    best_thr = min(16, ncpu)
    legs = [(2, best_thr)] if quick else [(2, best_thr), (2, min(8, ncpu)), (8, best_thr)]
    out = []
    for bp, thr in legs:
        out.append({"batch": bp, "threads": thr, "pairs_per_s": round(_cpu_leg(O, height, width, bp, thr), 4)})
    phys = physical_cores()
    if not quick and phys > best_thr:
        # BASELINE.md section 3 asks for the all-physical-cores figure: ONE timed step after one warm-up (bounded: this graph gets
        # SLOWER with more threads, so the leg is a record, not a contender -- ~30 s at 0.07 pairs/s on 128 cores)
        out.append({"batch": 2, "threads": phys, "pairs_per_s": round(_cpu_leg(O, height, width, 2, phys, warm=1, timed=1), 4),
                    "note": "all physical cores, 1 warm-up + 1 timed step"})
    top = max(out, key=lambda r: r["pairs_per_s"])
    return {"value": top["pairs_per_s"], "unit": "image-pairs/s", "cores": top["threads"], "kind": "port",
            "host_cpus": ncpu, "physical_cores": phys, "cpu_model": cpu_model(), "legs": out,
            "sample": f"oracle train step (fwd+loss+bwd+Adam) at {height}x{width}: 2 warm-up + 5 timed steps per leg, median step "
                      f"time; legs = (batch, torch threads) {[(r['batch'], r['threads']) for r in out]}; value = the fastest "
                      f"leg (batch {top['batch']}, {top['threads']} threads; {ncpu} host CPUs visible, more threads are slower)"}


def cpu_forward_leg(height, width, pairs=2, threads=16):
    """forward-only CPU baseline for BASELINE configs[1] + the checker's outputs: the oracle's eval forward on the first ``pairs``
    pairs of the forward-only leg's batch (robust-mask parity mode), timed once after one warm-up -> (pairs/s, flow, occ)"""
    from oracle import irr_pwc_oracle as O
    torch.set_num_threads(min(threads, os.cpu_count() or threads))
    P = O.synthetic_params(0)
    b = O.synthetic_batch(pairs, height, width, 4321)
    with torch.no_grad():
        O.irr_pwc_forward(P, b["input1"][:1], b["input2"][:1], False, mask_threshold=0.9999)
        t0 = time.perf_counter()
        out = O.irr_pwc_forward(P, b["input1"], b["input2"], False, mask_threshold=0.9999)
        dt = time.perf_counter() - t0
    return pairs / dt, out["flow"], out["occ"], P, b


def launch_ranks(n):
    """``python bench.py --gpus N`` from a cold shell: start N ranks with torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never re-execs), pass their output through, return their exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC for RCCL (see the task environment notes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prealloc-gb", type=float, default=-1.0,
                    help="device memory reserved in the caching allocator before the first step (one block, split on demand), so that "
                         "no hipMalloc lands in the timed steps; default: 30 %% of the device (86 GB on MI355X), 0 = off")
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=448)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="one CPU leg (batch 2) instead of three")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-every", type=int, default=4,
                    help="per-launch HIP events (roofline object) on every N-th step of the timed region, starting with its first "
                         "(an event pair per launch costs ~8 us of kernel overlap: 1.6 %% of a step when every step is instrumented)")
    ap.add_argument("--no-async-wgrad", action="store_true", help="keep weight gradients on the main stream")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 448x1024 leg after the headline timing")
    ap.add_argument("--harness", choices=["own", "reference"], default="own",
                    help="own: TrainStep + GradArena + FusedAdam (headline); reference: the reference's literal loop as the headline")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip reference_harness / forward_only / single_stream")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE={world} of the launcher")
    backend = os.environ.get("IRR_DDP_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # IRR_DDP_SINGLE_RANK=1 (test switch, irr_amd.ddp.collectives_on): a process group of ONE rank with every collective of the
    # data-parallel step really issued -- the RCCL transport and its stream choreography on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("IRR_DDP_SINGLE_RANK"))
    # stdout carries ONE JSON line (rank 0) and nothing else.  Libraries write there too: RCCL prints a five-line version banner per
    # communicator with NCCL_DEBUG=VERSION (set on the GPU boxes; C stdio, flushed at exit, i.e. BEHIND the JSON line; NCCL_DEBUG_FILE
    # does not move it), gloo its "connected to n peer ranks" lines.  So file descriptor 1 is pointed at stderr for the whole process and
    # the JSON line goes to a saved duplicate of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import irr_amd
    from irr_amd import build as B_
    from irr_amd import conv as C
    from irr_amd import ddp, harness
    from irr_amd.train import ModelAndLoss, TrainStep
    from irr_amd.optim import FusedAdam

    # the loaded library must be the one built from the sources in the tree: the stamp next to it is written by
    # irr_amd.build only after a link of exactly these sources
    if not os.environ.get("IRR_HIP_LIB") and B_.built_hash() != B_.source_hash():
        raise SystemExit(f"irr_amd/lib/libirr_hip.so was built from sources {B_.built_hash() or '?'}, the tree is "
                         f"{B_.source_hash()}: run `python -m irr_amd.build`")

    pre_gb = a.prealloc_gb if a.prealloc_gb >= 0 else 0.30 * torch.cuda.get_device_properties(device).total_memory / 1e9
    if backend != "nccl" and world > 1:
        pre_gb /= world                                      # the ranks share a device
    if pre_gb > 0:
        # a step peaks at 39 GB allocated / 69 GB reserved (bs32, 384x448): the pool would otherwise still grow by a few
        # segments during the first timed steps
        blk = torch.empty(int(pre_gb * 1e9), dtype=torch.uint8, device=device)
        del blk

    def new_model(batch_pairs):
        torch.manual_seed(0)                                 # same MSRA init on every rank
        return irr_amd.PWCNet(types.SimpleNamespace(batch_size=batch_pairs, model_div_flow=0.05)).to(device).train()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def own_step_factory(model, arena, opt):
        def make(batch_pairs):
            args = types.SimpleNamespace(batch_size=batch_pairs, model_div_flow=0.05)
            loss = irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args, reduce_fn=ddp.reduce_losses() if dist_on else None).train()
            mal = ModelAndLoss(args, model, loss).train()
            # the reference's per-step NaN assertion is ON: asserted before the optimizer step (TrainStep docstring);
            # IRR_BENCH_NANCHECK=before_backward: the reference's exact placement (A/B), =off: diagnostic
            nc = os.environ.get("IRR_BENCH_NANCHECK", "before_step")
            return TrainStep(mal, opt, grad_sync=arena.sync,
                             check_nan=False if (nc == "off" or os.environ.get("IRR_BENCH_NO_NANCHECK")) else nc)
        return make

    def reference_step_factory(model):
        """the reference's loop, literally (runtime.py:158-189); optimizer as configuration.py:488-573 builds it from
        scripts/IRR-PWC_flyingChairsOcc.sh:29-31"""
        import math
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=4e-4)

        def make(batch_pairs):
            args = types.SimpleNamespace(batch_size=batch_pairs, model_div_flow=0.05)
            mal = ModelAndLoss(args, model, irr_amd.MultiScaleEPE_PWC_Bi_Occ_upsample(args).train()).train()

            def step(example_dict):
                for key, t in example_dict.items():
                    t.requires_grad_("input" in key)
                optimizer.zero_grad()
                loss_dict, output_dict = mal(example_dict)
                training_loss = loss_dict["total_loss"]
                assert not math.isnan(training_loss.item()), "training_loss is NaN"
                training_loss.backward()
                optimizer.step()
                return loss_dict, output_dict, batch_pairs
            return step
        return make

    def run(make_step, batch_pairs, height, width, steps, warmup, timed_kernels):
        """W untimed + exactly K timed train steps of one workload -> dict(dt, value, loss, routing, roofline)"""
        step = make_step(batch_pairs)
        batches = [synthetic_batch(batch_pairs, height, width, 1234 + rank + int(os.environ.get("IRR_BENCH_SEED_OFFSET", "0")) + 1000 * i, device) for i in range(NBATCHES)]   # (offset: diagnosis switch -- another rank's batches in a single process)
        marks = [] if os.environ.get("IRR_BENCH_STEPTIMES") else None      # diagnostic: per-step GPU time (events, no extra syncs)

        def mark():
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((time.perf_counter(), ev))

        for i in range(warmup):
            mark()
            step(batches[i % NBATCHES])
        seg0 = torch.cuda.memory_stats(device).get("segment.all.allocated", 0) if marks is not None else 0
        timer = None
        if timed_kernels:
            timer = C.KernelTimer()
            C.TIMER = timer
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            mark()
            if timer is not None:
                timer.begin_step(i % max(1, a.timer_every) == 0)
            ld, _, _ = step(batches[(warmup + i) % NBATCHES])
        mark()
        barrier()
        dt = time.perf_counter() - t0
        C.TIMER = None
        if marks is not None and rank == 0:
            print("per-step ms (gpu / cpu issue), first %d are warm-up: " % warmup +
                  " ".join("%.1f/%.1f" % (m0[1].elapsed_time(m1[1]), 1e3 * (m1[0] - m0[0])) for m0, m1 in zip(marks[:-1], marks[1:])),
                  file=sys.stderr)
            st = torch.cuda.memory_stats(device)
            print("allocator: segments created in the timed steps %d (total %d), retries %d, peak allocated %.1f GB, reserved %.1f GB"
                  % (st.get("segment.all.allocated", 0) - seg0, st.get("segment.all.allocated", 0), st.get("num_alloc_retries", 0),
                     torch.cuda.max_memory_allocated(device) / 1e9, torch.cuda.max_memory_reserved(device) / 1e9), file=sys.stderr)
        # what one step was ROUTED to (launch counters of irr_amd.conv), taken on one extra step outside the timed region
        C.LAUNCHES.clear()
        step(batches[0])
        torch.cuda.synchronize()
        routing = dict(C.LAUNCHES)
        spread = None
        if dist_on:
            allt = torch.zeros(world, device=device, dtype=torch.float64)      # every rank fills its slot: SUM == gather
            allt[rank] = dt
            dist.all_reduce(allt, op=dist.ReduceOp.SUM)
            per_rank = [float(t) for t in allt.tolist()]
            dt = max(per_rank)                                                 # MAX over ranks
            spread = {"per_rank_ms_per_step": [round(t / steps * 1e3, 3) for t in per_rank]}
        res = {"dt": dt, "value": batch_pairs * world * steps / dt, "routing": routing, "spread": spread,
               "loss": {k: float(v.detach()) for k, v in ld.items()},
               "roofline": roofline(timer, timer.steps, height, width, batch_pairs) if (timer is not None and rank == 0) else None}
        del batches, step
        return res

    def roofline(timer, steps, height, width, batch_pairs):
        summ = timer.summary()
        if not summ:
            return None
        var, st = max(summ.items(), key=lambda kv: kv[1]["seconds"])
        ach = st["flops"] / st["seconds"] / 1e12
        tot_f = sum(s["flops"] for s in summ.values())
        tot_s = sum(s["seconds"] for s in summ.values())
        kname, peak, bound = kernel_name(var)
        mfma_note = ("algorithmic fp32 FLOPs (2*MACs); the fp16x2 form issues 3 fp16 MFMA products per fp32 product, "
                     "so its roof is the dense fp16 MFMA peak 2500 / 3 = 833.3 TFLOP/s" if peak == H2_PEAK_TFLOPS else
                     "algorithmic fp32 FLOPs (2*MACs); the bf16x3 form issues 6 bf16 MFMA products per fp32 product, "
                     "so its roof is the dense bf16 MFMA peak 2500 / 6 = 416.7 TFLOP/s" if peak != FP32_MFMA_PEAK_TFLOPS
                     else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)")
        excl = ({"achieved": round(st["fwd_flops"] / st["fwd_seconds"] / 1e12, 2),
                 "frac": round(st["fwd_flops"] / st["fwd_seconds"] / 1e12 / peak, 4),
                 "launches": st["fwd_calls"],
                 "avg_launch_us": round(st["fwd_seconds"] / st["fwd_calls"] * 1e6, 2)} if st["fwd_calls"] else None)
        if bound == "hbm":
            # the streaming 32-channel kernel: priced against HBM with the ALGORITHMIC bytes of its launches (input map
            # + output map + every residual / accumulate / mask / second-output map, each once); the MFMA view stays as a side key
            gbps = st["bytes"] / st["seconds"] / 1e9
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(gbps, 1), "peak": HBM_PEAK_TBPS * 1e3, "unit": "GB/s",
                    "frac": round(gbps / (HBM_PEAK_TBPS * 1e3), 4), "traffic": None,
                    "bytes_per_launch": st["bytes"] / st["calls"],
                    "peak_note": "algorithmic bytes per launch (every operand / result map once) / HIP-event duration against 8 TB/s of HBM3E",
                    "mfma_view": {"achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                                  "note": mfma_note},
                    "exclusive": ({"achieved": round(st["fwd_bytes"] / st["fwd_seconds"] / 1e9, 1),
                                   "frac": round(st["fwd_bytes"] / st["fwd_seconds"] / 1e9 / (HBM_PEAK_TBPS * 1e3), 4),
                                   "launches": st["fwd_calls"], "avg_launch_us": round(st["fwd_seconds"] / st["fwd_calls"] * 1e6, 2)}
                                  if st["fwd_calls"] else None)}
        else:
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None, "peak_note": mfma_note, "exclusive": excl}
        roof.update({
                "launches": st["calls"], "avg_launch_us": round(st["seconds"] / st["calls"] * 1e6, 2),
                "flop_per_launch": st["flops"] / st["calls"],
                "concurrent_lanes": 1 if a.no_async_wgrad else 2,
                "timed_steps": steps,
                "timing": f"HIP events on the launch stream around every conv launch of {steps} steps of the timed region (every "
                          f"{max(1, a.timer_every)}-th step, starting with the first); kernel = the template instantiation as rocprofv3 "
                          f"prints it (profiles/*_kernel_stats.txt)",
                "note": ("durations include time-sharing the chip with the asynchronous weight-gradient lane "
                         "(second HIP stream) during backward; 'exclusive' = the forward-pass launches of the same "
                         "kernel, which run alone") if not a.no_async_wgrad else "single stream",
                "all_conv_fwd_dgrad": {"achieved": round(tot_f / tot_s / 1e12, 2), "seconds_per_step": round(tot_s / steps, 5),
                                       "flops_per_step": tot_f / steps},
                "by_kernel_all": {kernel_name(v)[0]: {"launches": s_["calls"], "achieved": round(s_["flops"] / s_["seconds"] / 1e12, 2),
                                                      "avg_launch_us": round(s_["seconds"] / s_["calls"] * 1e6, 2),
                                                      "algorithmic_GBps": round(s_["bytes"] / s_["seconds"] / 1e9, 1)}
                                  for v, s_ in summ.items()},
                "by_kernel": {kernel_name(v)[0]: {"launches": s_["calls"], "ms_per_step": round(s_["seconds"] / steps * 1e3, 2),
                                                  "achieved": round(s_["flops"] / s_["seconds"] / 1e12, 1)}
                              for v, s_ in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"])[:6]}})
        # HBM bytes per launch of that kernel from the committed PMC passes (profiles/hbm_traffic*.json: separate --pmc FETCH_SIZE /
        # WRITE_SIZE runs of THIS command, FETCH x2 gfx950 correction).  PMC counters cannot be collected from inside the process,
        # so the bytes are only reported when the json was measured on a library built from the very sources that are loaded now.
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILES[(height, width, batch_pairs)])))
            k = roof["kernel"]
            if tr.get("_source_hash") != B_.source_hash() or B_.built_hash() != B_.source_hash():
                roof["traffic_source"] = (f"stale: {TRAFFIC_FILES[(height, width, batch_pairs)]} was measured on sources "
                                          f"{tr.get('_source_hash')}, the library is built from {B_.source_hash()}")
            elif k in tr:
                roof["traffic"] = round((tr[k]["fetch_MB_per_launch_corrected"] + tr[k]["write_MB_per_launch"]) * 1e6)
                roof["traffic_source"] = (f"profiles/{TRAFFIC_FILES[(height, width, batch_pairs)]} (rocprofv3 PMC passes over this "
                                          f"command, bytes per launch, sources {tr['_source_hash']})")
        except Exception:
            pass
        return roof

    def workload_name(cfg, batch_pairs, height, width):
        return (f"BASELINE {cfg}: IRR_PWC train step (fwd + MultiScaleEPE_PWC_Bi_Occ_upsample + bwd + Adam), "
                f"synthetic {height}x{width}, {batch_pairs} pairs per GPU")

    model = new_model(a.batch)
    ddp.broadcast_params(model)
    arena = opt = None
    if a.harness == "own":
        arena = ddp.GradArena(model.named_parameters())
        if not a.no_async_wgrad:
            arena.enable_async_wgrad()
        elif not os.environ.get("IRR_BENCH_AUTOGRAD_WGRAD"):     # single stream: still accumulate straight into the arena
            arena.enable_direct_wgrad()
        opt = FusedAdam(model, arena, lr=1e-4, weight_decay=4e-4)
        make_step = own_step_factory(model, arena, opt)
    else:
        if world > 1:
            raise SystemExit("--harness reference is the reference's single-process loop (the reference has no data parallelism)")
        make_step = reference_step_factory(model)

    head = run(make_step, a.batch, a.height, a.width, a.steps, a.warmup, not a.no_kernel_timer)
    # rank 0's bucket schedule of the last step: (bucket, where it was started, ms since zero_grad)
    launch_rep = arena.launch_report() if (arena is not None and dist_on) else None
    launch_log = launch_rep["bucket_launches"] if launch_rep is not None else None
    second = None
    if not a.no_secondary and (a.height, a.width) == (384, 448):
        second = run(make_step, SECONDARY[0], SECONDARY[1], SECONDARY[2], SECONDARY_STEPS, 2, not a.no_kernel_timer)

    extra = {}
    if world == 1 and not a.no_extra_legs and not a.no_kernel_timer:
        # what the per-launch HIP events of the timed region cost (~1 000 event pairs per step): 5 more steps without them
        nt = run(make_step, a.batch, a.height, a.width, 5, 0, False)
        extra["without_kernel_timer"] = {"value": round(nt["value"], 3), "ms_per_step": round(nt["dt"] / 5 * 1e3, 3), "steps": 5,
                                         "note": "same process, same step, KernelTimer off (the headline's timed region records two "
                                                 "HIP events around every conv launch for the roofline object)"}
    if world == 1 and not a.no_extra_legs and a.harness == "own":
        # what the reference's legacy `requires_grad_(True)` on the INPUT IMAGES costs (runtime.py:158-162; TrainStep(input_grads=)): 5 steps
        # of the same workload whose backward does not produce the two image gradients nobody reads.  Information, not the headline.
        def make_noig(batch_pairs, _m=make_step):
            st = _m(batch_pairs)
            st.input_grads = False
            return st
        ng = run(make_noig, a.batch, a.height, a.width, 5, 1, False)
        extra["without_input_grads"] = {"value": round(ng["value"], 3), "unit": "image-pairs/s", "ms_per_step": round(ng["dt"] / 5 * 1e3, 3),
                                        "steps": 5, "warmup": 1, "loss": ng["loss"],
                                        "note": "TrainStep(input_grads=False): the input images are not marked requires_grad (the reference marks "
                                                "them, runtime.py:158-162, and so does the headline); same losses, same parameter gradients, no "
                                                "d loss / d image"}
    if world == 1 and not a.no_extra_legs and a.harness == "own" and C.MATH == "h2":
        # the range-free fall-back arithmetic beside the headline: 5 steps of the same workload, same process, with every
        # split-operand conv on the bf16x3 form (three bf16 pieces, six products: 24 significant bits whatever the operand range)
        C.set_math("x3")
        try:
            x3r = run(make_step, a.batch, a.height, a.width, 5, 1, False)
        finally:
            C.set_math("h2")
        extra["conv_math_x3"] = {"value": round(x3r["value"], 3), "unit": "image-pairs/s", "ms_per_step": round(x3r["dt"] / 5 * 1e3, 3),
                                 "steps": 5, "warmup": 1, "loss": x3r["loss"],
                                 "note": "IRR_CONV_MATH=x3 / conv.set_math('x3'): bf16x3 split operands everywhere (no operand scaling, "
                                         "no amax slots); same model, same batches"}
        # ... and the operand-exact arithmetic: every conv on the fp32 MFMA (157.3 TFLOP/s peak), the third of the three routes side by
        # side in the driver's line (VERDICT r5 missing #5)
        C.set_math("f32")
        try:
            f32r = run(make_step, a.batch, a.height, a.width, 3, 1, False)
        finally:
            C.set_math("h2")
        extra["conv_math_f32"] = {"value": round(f32r["value"], 3), "unit": "image-pairs/s", "ms_per_step": round(f32r["dt"] / 3 * 1e3, 3),
                                  "steps": 3, "warmup": 1, "loss": f32r["loss"],
                                  "note": "IRR_CONV_MATH=f32 / conv.set_math('f32'): v_mfma_f32_32x32x2_f32 everywhere (fp32 operands, "
                                          "1/16 of the 16-bit matrix rate); same model, same batches"}
    if world == 1 and not a.no_extra_legs and a.harness == "own" and not a.no_async_wgrad and not a.no_kernel_timer:
        # (1) kernel quality without lane time-sharing: 3 steps with the weight gradients on the main stream, same process
        arena.disable_async_wgrad()
        arena.enable_direct_wgrad()
        ss = run(make_step, a.batch, a.height, a.width, 3, 1, True)
        arena.disable_async_wgrad()
        if head["roofline"] is not None and ss["roofline"] is not None:
            kn = head["roofline"]["kernel"]
            k_ss = ss["roofline"]["by_kernel_all"].get(kn)
            if k_ss is not None:
                ach_ss = k_ss["algorithmic_GBps"] if head["roofline"]["bound"] == "hbm" else k_ss["achieved"]
                head["roofline"]["single_stream"] = {
                    "achieved": ach_ss, "frac": round(ach_ss / head["roofline"]["peak"], 4),
                    "avg_launch_us": k_ss["avg_launch_us"], "launches": k_ss["launches"], "ms_per_step": round(ss["dt"] / 3 * 1e3, 3),
                    "note": "3 timed steps of the same workload with the weight-gradient launches on the main stream "
                            "(GradArena.enable_direct_wgrad): no second kernel shares the chip"}
        # (2) the reference's own loop around a FRESH model (the headline's model / arena / optimizer are dropped first)
        del make_step, opt, arena, model
        ref_steps = max(3, min(a.steps, 8))
        legs = {}
        for name, auto in (("auto", True), ("plain_autograd", False)):
            harness.set_enabled(auto)
            m2 = new_model(a.batch)
            r = run(reference_step_factory(m2), a.batch, a.height, a.width, ref_steps, 3, False)
            legs[name] = {"value": round(r["value"], 3), "ms_per_step": round(r["dt"] / ref_steps * 1e3, 3), "loss": r["loss"],
                          "launches_per_step": r["routing"], "installed": harness.installed(m2)}
            harness.uninstall(m2)
            del m2, r
        harness.set_enabled(True)
        extra["reference_harness"] = {
            "value": legs["auto"]["value"], "unit": "image-pairs/s", "ms_per_step": legs["auto"]["ms_per_step"],
            "steps": ref_steps, "warmup": 3, "loss": legs["auto"]["loss"], "launches_per_step": legs["auto"]["launches_per_step"],
            "loop": "runtime.py:158-189 literally: optimizer.zero_grad() (set_to_none), forward, .item() NaN assertion before "
                    "backward(), torch.optim.Adam(lr 1e-4, weight_decay 4e-4).step(); no GradArena / FusedAdam / TrainStep",
            "route": "the model installs gradient arena + weight-gradient lane itself (irr_amd/harness.py), joined at the end of "
                     "backward by an autograd final callback; packed weights refreshed by one batched launch per step",
            "arena_installed_by_model": legs["auto"]["installed"],
            "plain_autograd": {"value": legs["plain_autograd"]["value"], "ms_per_step": legs["plain_autograd"]["ms_per_step"],
                               "launches_per_step": legs["plain_autograd"]["launches_per_step"],
                               "note": "IRR_AUTO_LANE=0: per-use gradient tensors, autograd accumulation, weight gradients on "
                                       "the critical path"}}
        # (3) BASELINE configs[1]: forward only, 8 pairs of 384x448, eval mode
        if (a.height, a.width) == (384, 448):
            m3 = new_model(8).eval()
            fb = synthetic_batch(8, a.height, a.width, 4321, device)
            with torch.no_grad():
                for _ in range(3):
                    m3(fb)
                torch.cuda.synchronize()
                n_f = 20
                t0 = time.perf_counter()
                for _ in range(n_f):
                    fo = m3(fb)
                torch.cuda.synchronize()
                dtf = time.perf_counter() - t0
            extra["forward_only"] = {"workload": "BASELINE configs[1]: IRR_PWC forward only (eval), synthetic 384x448, 8 pairs, 1 GPU",
                                     "value": round(8 * n_f / dtf, 2), "unit": "image-pairs/s", "ms_per_forward": round(dtf / n_f * 1e3, 3),
                                     "steps": n_f, "warmup": 3}
            extra["_fwd"] = (m3, fo)
    if rank == 0:
        gf = CONV_GFLOP_PER_PAIR.get((a.height, a.width))
        value = head["value"]
        own = a.harness == "own"
        out = {"metric": "image-pairs/sec fwd+bwd IRR-PWC 384x448 bs32", "value": round(value, 3), "unit": "image-pairs/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(head["dt"] / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"h2": "f32 (fp16x2 split operands)", "x3": "f32 (bf16x3 split operands)"}.get(C.MATH, "f32"), "data": "synthetic",
               "config": {"workload": workload_name("configs[2]" if (a.height, a.width) == (384, 448) else "(other crop)",
                                                    a.batch, a.height, a.width).replace("synthetic", "FlyingChairsOcc-shaped synthetic"),
                          "pairs_per_gpu": a.batch, "global_batch": a.batch * world, "height": a.height, "width": a.width,
                          "parallelism": f"dp{world}", "weights": "MSRA init, torch.manual_seed(0)",
                          "transport": "rccl" if backend == "nccl" else backend,
                          "harness": ("TrainStep + GradArena + FusedAdam (irr_amd/train.py)" if own else
                                      "the reference's loop (runtime.py:158-189) with torch.optim.Adam"),
                          "batches": f"{NBATCHES} distinct synthetic batches resident in HBM, fed round-robin",
                          "nan_check": (os.environ.get("IRR_BENCH_NANCHECK", "before_step") + " (every step: the reference's assertion on "
                                        "the training loss, runtime.py:182-183; before_step = asserted before the optimizer step "
                                        "from a pinned host copy, no pipeline drain between forward and backward)") if own else
                                       "before_backward (.item(), as the reference)"},
               "loss": head["loss"],
               "conv_math": C.MATH,
               "x3s_h2": bool(C.X3S_H2),                   # the streaming 32-channel kernel's fp16x2 form (default since round 5; IRR_X3S_H2=0: bf16x3)
               "x3s_mask_bits": bool(C.X3S_BITS),          # its LeakyReLU' masks as bits (ABI 9; IRR_X3S_BITS=0: fp32 activations)
               "branch_streams": (f"occlusion branch of levels < {os.environ.get('IRR_BRANCH_LEVELS', '4')} on a second HIP stream"
                                  if os.environ.get("IRR_BRANCH_STREAMS", "1") != "0" else "off"),
               "conv_math_note": {"h2": "fp32 tensors in HBM, fp32 accumulation and results; the MFMA convs split every operand, scaled by a "
                                        "power of two from max|.| of its tensor, into two fp16 pieces and accumulate three piece products "
                                        "(v_mfma_f32_32x32x16_f16).  The activation-side low piece is stored x 2^11 (its partner, the "
                                        "weight-side high piece, x 2^-11 in registers): 22-23 significant bits for every ELEMENT within "
                                        "2^29 (5e8 : 1) of its tensor's maximum, absolute 2^-36 of the scaled range below -- regional "
                                        "tests in tests/test_h2_gpu.py (quiet samples / rows / channels at 1e-5 ... 1e-7); the weight "
                                        "gradient's gy-role operand keeps the plain pair under one scale per channel (round 6).  'conv_math_x3' = the same step on "
                                        "bf16x3 (range-free, 24 bits); IRR_CONV_MATH=f32: fp32 MFMA",
                                  "x3": "fp32 tensors in HBM; the MFMA convs split every operand into three bf16 pieces and accumulate six "
                                        "piece products in fp32", "f32": "fp32 MFMA everywhere"}.get(C.MATH),
               "launches_per_step": head["routing"],
               "step_conv_tflops": round(value / world * gf * 1e9 / 1e12, 1) if gf else None,
               # whole-step conv rate over the fp32-MFMA peak (157.3): above 1 because the x3 family runs fp32 products on the bf16 pipe
               "step_conv_vs_fp32_mfma_peak": round(value / world * gf * 1e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4) if gf else None,
               "roofline": head["roofline"]}
        if out["roofline"] is not None:
            out["roofline"].pop("by_kernel_all", None)
        if head["spread"] is not None:
            out["ranks"] = head["spread"]
            out["ranks"]["bucket_launches_last_step"] = launch_log      # rank 0: (bucket, "backward" | "sync", ms since zero_grad)
            if launch_rep is not None:
                # ms since zero_grad (host clock) at which the cost-volume gradient kernels of levels 4 .. 0 were issued: bucket 0's
                # all-reduce is enqueued before the first of them (it runs under all five), bucket 1's around the last (it is reduced
                # under the pyramid's backward)
                out["ranks"]["corr_backward_launches_ms"] = launch_rep["corr_backward_launches_ms"]
        if second is not None:
            gf2 = CONV_GFLOP_PER_PAIR[(SECONDARY[1], SECONDARY[2])]
            if second["roofline"] is not None:
                second["roofline"].pop("by_kernel_all", None)
            out["secondary"] = {"workload": workload_name("configs[4] per-GPU share", *SECONDARY).replace("synthetic", "Sintel-shaped synthetic"),
                                "metric": "image-pairs/sec fwd+bwd IRR-PWC 448x1024 bs8", "value": round(second["value"], 3),
                                "unit": "image-pairs/s", "steps": SECONDARY_STEPS, "warmup": 2,
                                "ms_per_step": round(second["dt"] / SECONDARY_STEPS * 1e3, 3), "loss": second["loss"],
                                "step_conv_tflops": round(second["value"] / world * gf2 * 1e9 / 1e12, 1),
                                "launches_per_step": second["routing"], "roofline": second["roofline"]}
        fwd = extra.pop("_fwd", None)
        out.update(extra)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.height, a.width, quick=a.quick_cpu_baseline)
            if fwd is not None:
                # configs[1]'s "EPE check": the oracle's eval forward on the first two pairs of the forward-only batch, timed as
                # that leg's CPU baseline and compared with the HIP outputs (robust-mask parity mode on both sides)
                m3, _ = fwd
                pps, flow_ref, occ_ref, P, cb = cpu_forward_leg(a.height, a.width)
                m3.load_state_dict(P)
                m3.mask_threshold = 0.9999
                with torch.no_grad():
                    got = m3({"input1": cb["input1"].to(device), "input2": cb["input2"].to(device)})
                epe = torch.norm(got["flow"].cpu() - flow_ref, dim=1).mean().item()
                out["forward_only"]["epe_vs_oracle_px"] = float(f"{epe:.3e}")
                out["forward_only"]["occ_logit_mad_vs_oracle"] = float(f"{(got['occ'].cpu() - occ_ref).abs().mean().item():.3e}")
                out["forward_only"]["epe_check"] = ("2 pairs of 384x448, oracle weights (synthetic_params(0)), mask threshold 0.9999 "
                                                    "on both sides; bar 1e-4 px (SURVEY 8(c))")
                out["forward_only"]["cpu_baseline"] = {"value": round(pps, 4), "unit": "image-pairs/s", "cores": min(16, os.cpu_count() or 16),
                                                       "kind": "port", "sample": "oracle eval forward, 2 pairs of 384x448, one timed pass"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
