#!/usr/bin/env python3
"""Benchmark of the SRGAN-ST training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload srgan|srresnet|srgan_vgg] [--hr 96|192] [--batch B]

Default workload = BASELINE.json's metric: the "G+D+ST-loss step" (reference train.py:116-164 with D_UPDATE_INTERVAL = 1):
SRGAN x4, 96 px HR crops, B = 16 per GPU, generator step (adversarial 1e-3 + pixel MSE + structure tensor / 3, D frozen but in
train mode) followed by the discriminator step on gt and sr.detach(), both with Adam.  `value`, `ms_per_step`, `roofline` and
`cpu_baseline` all describe THAT step.  The SRResNet step of BASELINE configs[1] (warmup.py:74-96, G only) is measured in the same
run under the same protocol and reported as the secondary key `srresnet_step`.
`--workload srgan_vgg` = configs[2] with the VGG19 content term (seeded-random VGG19: the ImageNet weights are a network fetch).
Prints ONE JSON line (rank 0).  Inputs are synthetic DIV2K-shaped tensors already resident in HBM.
Exit status: 0 only when every measured leg finished; a stalled or failed leg still prints the line (with an "error" key)
and then exits 3.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.environ.get("SST_PKG_ROOT") or os.path.join(ROOT, "srgan-st_amd")):     # SST_PKG_ROOT: dev A/B of two package trees
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

G_FWD_MAC_PER_IMG = 1277.67e6      # BASELINE.md section 2 (HR 96)
D_FWD_MAC_PER_IMG = 884.15e6
VGG_FWD_MAC_PER_IMG = 3583.18e6
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0
# kernels of the 3-channel first discriminator layer (model.py:32): on the matrix cores, but one pass over a 37.7 MB tensor each
STREAMING_3CH = ("conv3_to3_kernel", "wgrad_k3c3_mfma_kernel", "conv3_c3in_mfma_kernel", "conv3_c3in_kernel", "wgrad_k3c3_kernel")

WORKLOADS = {
    # name -> (description, MACs per image per step as multiples of (G fwd, D fwd, VGG fwd))
    "srgan": ("srgan_x4_hr96_b16_G+D+ST-loss step: adv+mse+st, D updated every step (train.py:116-164; BASELINE metric)", (3, 8, 0)),
    "srresnet": ("srresnet_x4_hr96_b16_mse+st (warmup.py:74-96; BASELINE configs[1])", (3, 0, 0)),
    "srgan_vgg": ("srgan_x4_hr96_b16_adv+vgg(random weights)+mse+st_D-every-step (BASELINE configs[2])", (3, 8, 3)),
}


def workload_name(workload, hr, B):
    return WORKLOADS[workload][0].replace("hr96", f"hr{hr}").replace("_b16_", f"_b{B}_")


def flop_per_image(workload, hr, d_sr_reused=False):
    """FLOP the step EXECUTES per image.  d_sr_reused: the discriminator step's D(sr.detach()) forward is shared with the generator
    step's D(sr) (engine.TrainEngine, KERNEL.REUSE_D_SR: bit-identical results) - one discriminator forward less than the reference runs."""
    g, d, v = WORKLOADS[workload][1]
    if d_sr_reused:
        d -= 1
    return 2.0 * (g * G_FWD_MAC_PER_IMG + d * D_FWD_MAC_PER_IMG + v * VGG_FWD_MAC_PER_IMG) * (hr / 96.0) ** 2


def synth_batch(B, hr, device, seed):
    """HR crops on the 1/255 grid + x1/4 LR (antialiased bicubic, re-quantised like dataset.py:27-28)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    gt = torch.randint(0, 256, (B, 3, hr, hr), generator=g, dtype=torch.uint8).float() / 255.0
    lr = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True, align_corners=False)
    lr = torch.round(lr * 255.0) / 255.0
    return gt.to(device), lr.to(device)


def build_engine(workload, device, use_graph, hr, share_d_sr=False):
    """share_d_sr=False (the headline): the iteration runs all three discriminator forwards of the reference's step (train.py:136,
    155, 158).  True: the engine's default schedule, in which the discriminator step works on the generator step's D(sr) pass
    (KERNEL.REUSE_D_SR: bit-identical results, one forward less) - reported next to the headline, never as the headline."""
    from srganst.config import Config
    from srganst.engine import TrainEngine, WarmupEngine
    from srganst.loss import MSELoss, StructureTensorLoss
    from srganst.model import Discriminator, Generator
    cfg = Config()
    cfg.DEVICE = str(device)
    cfg.DATA.GT_IMAGE_SIZE = hr
    torch.manual_seed(cfg.DATA.SEED)
    if workload == "srresnet":
        G = Generator(cfg).to(device).train()
        crits = {"Pixel": MSELoss(), "ST": StructureTensorLoss()}
        w = {"Pixel": 1.0, "ST": 1.0 / 3.0}
        return WarmupEngine(cfg, G, crits, w, use_graph=use_graph), cfg
    D = Discriminator(cfg).to(device).train()
    G = Generator(cfg).to(device).train()
    if workload == "srgan_vgg":
        from srganst.loss import ContentLossVGG
        cfg.add_g_criterion("ContentVGG", ContentLossVGG(cfg, allow_random=True), 1.0)     # seeded-random VGG19 (ImageNet weights are a fetch)
    cfg.add_g_criterion("Pixel", MSELoss(), 1.0)
    cfg.add_g_criterion("ST", StructureTensorLoss(), 1.0 / 3.0)
    cfg.SOLVER.D_UPDATE_INTERVAL = 1
    cfg.KERNEL.REUSE_D_SR = bool(share_d_sr)
    return TrainEngine(cfg, G, D, use_graph=use_graph), cfg


def _replay_time(calls, reps=20):
    """Seconds for one back-to-back pass over `calls` (relaunch closures), replayed from a hipGraph on the launch stream and
    bracketed by HIP events on that stream (GPU-bound: no host gaps inside the timed region)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for fn, _, _ in calls:
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for fn, _, _ in calls:
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def kernel_roofline(workload, device, hr, B):
    """Live HIP-event timing of the step's kernels: one eager step records every launch of a kernel family (same pointers,
    same shapes); each family is then replayed back-to-back from a hipGraph.  MFMA families are priced against the fp32
    MFMA peak (algorithmic FLOPs = 2*B*Ho*Wo*Cout*Cin*k*k per launch), HBM-bound families (structure-tensor loss, classifier
    GEMMs, Adam) against 8 TB/s (algorithmic bytes per launch).  The family with the largest time share
    is the headline `roofline`."""
    from srganst import ops
    eng, _ = build_engine(workload, device, use_graph=False, hr=hr)
    gt, lr = synth_batch(B, hr, device, 1)
    saved_overlap, ops.OVERLAP = ops.OVERLAP, False
    eng.step(gt, lr)
    torch.cuda.synchronize()
    ops.TRACE, ops.TRACE_HBM = {}, {}
    eng.step(gt, lr)
    torch.cuda.synchronize()
    trace, hbm = ops.TRACE, ops.TRACE_HBM
    ops.TRACE = ops.TRACE_HBM = None
    ops.OVERLAP = saved_overlap
    rows = {}
    for name, calls in trace.items():
        t = _replay_time(calls)
        flops = sum(f for _, f, _ in calls)
        if name.split("+")[0] in STREAMING_3CH:
            # 3x3 convs with a 3-channel side (the discriminator's first layer: forward, data-gradient, weight gradient) run on the
            # matrix cores but are bound by streaming the 64-channel tensor once: priced against HBM.  Algorithmic bytes = the
            # 64-channel tensor + the 3-channel tensor (B*H*W pixels = flops / (2 * 64 * 27))
            nbytes = flops / (2.0 * 64 * 27) * (64 + 3) * 4.0
            rows[name] = {"bound": "hbm", "launches_per_step": len(calls), "avg_launch_us": t / len(calls) * 1e6, "gbs": nbytes / t / 1e9,
                          "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "ms_per_step": t * 1e3, "bytes_per_launch": nbytes / len(calls),
                          "tflops": flops / t / 1e12}
            continue
        rows[name] = {"bound": "mfma", "launches_per_step": len(calls), "avg_launch_us": t / len(calls) * 1e6, "tflops": flops / t / 1e12,
                      "frac": flops / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, "ms_per_step": t * 1e3, "flop_per_launch": flops / len(calls)}
    for name, calls in hbm.items():
        t = _replay_time(calls)
        nbytes = sum(f for _, f, _ in calls)
        rows[name] = {"bound": "hbm", "launches_per_step": len(calls), "avg_launch_us": t / len(calls) * 1e6, "gbs": nbytes / t / 1e9,
                      "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "ms_per_step": t * 1e3, "bytes_per_launch": nbytes / len(calls)}
    eng.close()
    # One kernel template = one family: rocprofv3 lists the instantiations of a template separately (conv_pipe_kernel<1, 8>,
    # <1, 4>, <2, 8> ...: stride and tile width; conv_band_kernel<3, true> / <3, false> are already merged by the library's
    # label).  A family row = all launches of the template's instantiations: sum of their FLOPs (bytes) over the sum of their times.
    fams = {}
    for name, r_ in rows.items():
        fam = re.sub(r"<[^>]*>", "", name.split("+")[0]).replace("(grouped)", "")
        fams.setdefault(fam, []).append(name)
    for fam, members in fams.items():
        if len(members) < 2:
            continue
        t = sum(rows[m]["ms_per_step"] for m in members) * 1e-3
        n = sum(rows[m]["launches_per_step"] for m in members)
        if rows[members[0]]["bound"] == "mfma":
            fl = sum(rows[m]["flop_per_launch"] * rows[m]["launches_per_step"] for m in members)
            rows[fam] = {"bound": "mfma", "launches_per_step": n, "avg_launch_us": t / n * 1e6, "tflops": fl / t / 1e12,
                         "frac": fl / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, "ms_per_step": t * 1e3, "flop_per_launch": fl / n,
                         "instantiations": sorted(members)}
        else:
            nb = sum(rows[m]["bytes_per_launch"] * rows[m]["launches_per_step"] for m in members)
            rows[fam] = {"bound": "hbm", "launches_per_step": n, "avg_launch_us": t / n * 1e6, "gbs": nb / t / 1e9,
                         "frac": nb / t / 1e9 / PEAK_HBM_GBS, "ms_per_step": t * 1e3, "bytes_per_launch": nb / n,
                         "instantiations": sorted(members)}
    dom = max(rows, key=lambda k: rows[k]["ms_per_step"])
    r = rows[dom]
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # HBM bytes per launch from rocprofv3 --pmc passes
    if os.path.exists(tf):
        # keyed by the exact configuration the counters were collected on (workload, crop size, batch): a line for another shape
        # carries null, never another shape's bytes
        try:
            traffic = json.load(open(tf)).get(f"{workload}_hr{hr}_b{B}", {}).get(dom)
        except Exception:
            traffic = None
    if r["bound"] == "mfma":
        head = {"bound": "mfma", "kernel": dom, "achieved": r["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": r["frac"], "traffic": traffic, "flop_per_launch": r["flop_per_launch"]}
    else:
        head = {"bound": "hbm", "kernel": dom, "achieved": r["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": r["frac"],
                "traffic": traffic, "bytes_per_launch": r["bytes_per_launch"]}
    head.update({"avg_launch_us": r["avg_launch_us"], "launches_per_step": r["launches_per_step"], "kernels": rows})
    return head


def _host_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))     # the 1-GPU box's CPU share is 16 cores; oversubscribing it is far slower


def _time_cpu(fn, gt, lr, budget_s):
    fn(gt, lr)                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        fn(gt, lr)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 400:
            break
    return n, el


def cpu_baseline(workload, B, hr, budget_s=18.0):
    """The CPU oracle (plain-torch restatement of the reference step) timed on the host cores: a bounded sample of the SAME
    step as `value`, plus BASELINE configs[0] (SRResNet warm-up step, B = 4, pixel-L1 only: the reference's own CPU case)."""
    from oracle import model as om
    from oracle import steps as osteps
    torch.manual_seed(0)
    threads = _host_threads()
    torch.set_num_threads(threads)
    if workload == "srresnet":
        tr = osteps.OracleTrainer(om.init_generator_state(), criterions=(("Pixel", 1.0), ("ST", 1.0 / 3.0)))
        fn = tr.warmup_step
    elif workload == "srgan_vgg":
        from oracle import vgg as ovgg
        vsd = ovgg.init_vgg_state(0)
        d0 = om.init_discriminator_state(image_size=hr)
        tr = osteps.OracleTrainer(om.init_generator_state(), d0, criterions=(("Adversarial", 0.001), ("ContentVGG", 1.0), ("Pixel", 1.0),
                                                                             ("ST", 1.0 / 3.0)),
                                  d_update_interval=1, vgg=lambda sr, gt: ovgg.content_loss(vsd, sr, gt))
        fn = tr.train_step
    else:
        d0 = om.init_discriminator_state(image_size=hr)
        tr = osteps.OracleTrainer(om.init_generator_state(), d0, criterions=(("Adversarial", 0.001), ("Pixel", 1.0), ("ST", 1.0 / 3.0)),
                                  d_update_interval=1)
        fn = tr.train_step
    gt, lr = synth_batch(B, hr, "cpu", 1)
    n, el = _time_cpu(fn, gt, lr, budget_s)
    out = {"value": B * n / el, "unit": "HR images/s", "cores": threads, "kind": "port",
           "sample": f"{n} steps of the same B={B} {hr}px {workload} step through oracle/ (torch CPU eager, {threads} threads), {el:.1f} s"}
    # BASELINE configs[0]: warmup.py on CPU, 96->24 x4, batch 4, pixel-L1 only
    tr0 = osteps.OracleTrainer(om.init_generator_state(), criterions=(("Pixel", 1.0),), pixel_kind="l1")
    gt0, lr0 = synth_batch(4, 96, "cpu", 2)
    n0, el0 = _time_cpu(tr0.warmup_step, gt0, lr0, 5.0)
    out["configs0_srresnet_b4_l1"] = {"value": 4 * n0 / el0, "unit": "HR images/s", "cores": threads, "kind": "port",
                                      "sample": f"{n0} warm-up steps, B=4, 96 px, pixel-L1 only (BASELINE configs[0]), {el0:.1f} s"}
    return out


def timed_steps(eng, gt, lr, steps, warmup, world, device):
    """`warmup` untimed steps (>= 4: eager warm-ups + graph capture happen there), then exactly `steps` steps bracketed by
    barrier + synchronize on both sides; returns the elapsed seconds, MAX over ranks."""
    for _ in range(max(warmup, 4)):
        eng.step(gt, lr)
    gt, lr = eng.gt, eng.lr          # the batch now sits in the engine's static input buffers (what a loader fills by H2D copy):
    torch.cuda.synchronize()         # the timed steps read it there, no device-to-device copy per step
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(gt, lr)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    return el


STALL_EXIT = 3


def launch_ranks(n, argv, limit_s=1500.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, torchrun-style environment,
    rendezvous on 127.0.0.1) from a parent that never makes a GPU call and never re-execs, relay rank 0's JSON line, and return
    non-zero if any rank does (the other ranks are then ended by their exact PIDs)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)     # drain the pipe while rank 0 runs
    reader.start()
    rc, t0 = 0, time.time()
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print(f"bench.py: rank {r} exited with {c}; ending the other ranks", file=sys.stderr, flush=True)
        if live and (rc != 0 or time.time() - t0 > limit_s):
            if rc == 0:
                rc = STALL_EXIT
                print(f"bench.py: ranks {sorted(live)} still running after {limit_s:.0f} s; ending them", file=sys.stderr, flush=True)
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        if live:
            time.sleep(0.2)
    reader.join(10)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return rc


def secondary_leg(primary, workload, rank, world, device, args, B, steps=40, warmup=6, limit_s=240.0, share_d_sr=False, key=None):
    """A second workload under the same timing protocol (the configs[1] SRResNet step next to the headline G+D+ST step).
    Every rank runs a watchdog: if the leg is not done after limit_s, rank 0 still prints the primary line (with the reason in
    it), the reason goes to stderr and the process exits NON-ZERO - a stall is a failed run, not a result."""
    import threading
    from srganst import dist as sdist
    done = threading.Event()
    key = key or workload + "_step"

    def watchdog():
        if not done.wait(limit_s):
            msg = f"{key}: not finished after {limit_s:.0f} s (stalled GPU leg or collective)"
            if rank == 0:
                primary[key] = {"error": msg}
                print(json.dumps(primary), flush=True)
            print(f"bench.py: {msg}; exiting {STALL_EXIT}", file=sys.stderr, flush=True)
            os._exit(STALL_EXIT)

    threading.Thread(target=watchdog, daemon=True).start()
    res = None
    try:
        eng, _ = build_engine(workload, device, use_graph=not args.no_graph, hr=args.hr, share_d_sr=share_d_sr)
        if world > 1:
            sdist.broadcast_module(eng.G)
            if hasattr(eng, "D"):
                sdist.broadcast_module(eng.D)
        gt, lr = synth_batch(B, args.hr, device, seed=100 + rank)
        el = timed_steps(eng, gt, lr, steps, warmup, world, device)
        imgs = B * world * steps / el
        reused = bool(getattr(eng, "d_sr_reused", False))
        res = {"workload": workload_name(workload, args.hr, B), "value": imgs, "unit": "HR images/s", "ms_per_step": el / steps * 1e3,
               "steps": steps, "warmup": warmup, "n_gpus": world, "hip_graph": bool(eng.graph_active),
               "step_tflops": flop_per_image(workload, args.hr, reused) * imgs / 1e12}
        if share_d_sr:
            res["d_sr_forward"] = ("shared with the generator step's D(sr) pass: same input, same weights, deterministic kernels - results "
                                   "bit-identical to the three-pass step (tests/test_discriminator_gpu.py), one discriminator forward less"
                                   if reused else "run (sharing did not engage)")
        eng.close()
    except Exception as e:  # noqa: BLE001 - the primary line must still be printed; main() then exits non-zero
        res = {"error": f"{type(e).__name__}: {e}"}
        print(f"bench.py: {key} failed: {res['error']}", file=sys.stderr, flush=True)
    done.set()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="srgan", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--hr", type=int, default=96)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="run the big single weight gradients on a side stream (off: measured slower)")
    ap.add_argument("--no-overlap", action="store_true", help="(default) keep weight gradients on the main stream")
    ap.add_argument("--share-d-sr", action="store_true",
                    help="headline on the engine's default schedule (the discriminator step re-uses the generator step's D(sr) pass); "
                         "without it the headline runs all three discriminator forwards of the reference's step and the shared "
                         "schedule is reported as the secondary key srgan_shared_d_sr_step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", "--no-full-step", dest="no_secondary", action="store_true",
                    help="skip the secondary SRResNet (configs[1]) measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 code path with several "
                         "ranks sharing one GPU: ranks then map onto the visible devices modulo their count)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # this process never touches the GPU: it only starts and watches the ranks

    from srganst import _abi, dist as sdist, ops as _ops
    _abi.lib()                                    # fail loudly if the HIP extension is missing
    _ops.OVERLAP = bool(args.overlap) and not args.no_overlap
    if args.backend == "gloo":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    rank, local, world = sdist.init_from_env(args.backend)
    if world != args.gpus:
        # a launcher that started a different number of ranks than --gpus says would record one job under another's name
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (rank {rank}): refusing to measure", file=sys.stderr, flush=True)
        sys.exit(2)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    eng, cfg = build_engine(args.workload, device, use_graph=not args.no_graph, hr=args.hr, share_d_sr=args.share_d_sr)
    if world > 1:
        sdist.broadcast_module(eng.G)
        if hasattr(eng, "D"):
            sdist.broadcast_module(eng.D)
    B = args.batch
    gt, lr = synth_batch(B, args.hr, device, seed=100 + rank)      # distinct shard per rank, resident in HBM

    el = timed_steps(eng, gt, lr, args.steps, args.warmup, world, device)
    losses = {k: float(v) for k, v in eng.loss_values.items()}
    graph_active = bool(eng.graph_active)         # what actually happened, not the flag: a failed capture falls back to eager
    d_sr_reused = bool(getattr(eng, "d_sr_reused", False))
    eng.close()

    out = None
    failed = False
    comm = ({"backend": "rccl" if torch.distributed.get_backend() == "nccl" else torch.distributed.get_backend(),
             "ranks": torch.distributed.get_world_size()} if world > 1 else {"backend": None, "ranks": 1})
    if rank == 0:
        ms = el / args.steps * 1e3
        imgs = B * world * args.steps / el
        wl = workload_name(args.workload, args.hr, B)
        step_kind = {"srgan": "G+D+ST-loss step", "srresnet": "SRResNet G-only mse+ST step", "srgan_vgg": "G+D+VGG+ST-loss step"}[args.workload]
        out = {"metric": f"HR images/sec ({args.hr}px x4, B={B}/GPU) {step_kind}", "value": imgs, "unit": "HR images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "timed_region_s": el, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": wl, "global_batch": B * world, "hr": args.hr, "lr": args.hr // 4,
                          "parallelism": f"dp{world}", "comm": comm, "hip_graph": graph_active, "d_update_interval": 1 if args.workload != "srresnet" else None,
                          "d_sr_forward": ("shared with the generator step's D(sr) pass (same input and weights, bit-identical)"
                                           if d_sr_reused else ("run: all three discriminator forwards of train.py:136,155,158"
                                                                if args.workload != "srresnet" else None)),
                          "step_tflops": flop_per_image(args.workload, args.hr, d_sr_reused) * imgs / 1e12, "losses_last_step": losses}}
    if args.workload == "srgan" and not args.no_secondary:
        # BASELINE configs[1] (SRResNet, G only) in the same run, same protocol, fewer steps; at every N, so the driver's
        # scaling runs also exercise the generator-only gradient all-reduce.
        extra = secondary_leg(out, "srresnet", rank, world, device, args, B)
        failed = failed or (extra is not None and "error" in extra)
        if rank == 0:
            out["srresnet_step"] = extra
        if not args.share_d_sr:
            # the engine's default schedule (one discriminator forward less, bit-identical results) under the same protocol
            extra2 = secondary_leg(out, "srgan", rank, world, device, args, B, share_d_sr=True, key="srgan_shared_d_sr_step")
            failed = failed or (extra2 is not None and "error" in extra2)
            if rank == 0:
                out["srgan_shared_d_sr_step"] = extra2
    if rank == 0 and world == 1 and not args.no_roofline:
        out["roofline"] = kernel_roofline(args.workload, device, args.hr, B)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, B, args.hr)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if failed:
        sys.stdout.flush()
        os._exit(STALL_EXIT)                       # ranks may have diverged: no closing collective, and the run is a failure
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
