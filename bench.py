#!/usr/bin/env python3
"""Headline benchmark: utterances/sec, FastGRNN forward+backward, T=99 feat=32
hidden=128, bs=4096 per GPU (BASELINE.json `metric`), fp32, synthetic MFCC frames.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL);
batch shards data-parallel (weak scaling: 4096 utterances per GPU) and the parameter
gradients are all-reduced in one flattened bucket per step.  Rank 0 prints ONE JSON line.
A "step" = zero grads, forward over the whole [T,B,F] batch, backward with a dense
grad_hs ~ N(0,1), and (N > 1) the gradient all-reduce; inputs are resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T, F, H, B_PER_GPU = 99, 32, 128, 4096
# algorithmic work per utterance (dense, fp32)
FLOPS_FWD = T * (2 * F * H + 2 * H * H)          # 4 055 040
FLOPS_BWD = 2 * FLOPS_FWD                        # 8 110 080
BYTES_FWD = 4 * T * (F + H)                      # 63 360: read x, write hs
BYTES_BWD = 4 * T * (2 * H + 2 * F)              # 126 720: read grad_hs, hs, x; write d_x
PEAK_F32_TFLOPS = 157.3                          # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0                        # dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0                            # spec; ~6300 achievable
# bytes per utterance the autograd path actually moves when it saves ONE auxiliary tensor
# (kernel path 2): fwd reads x, writes hs + pre; bwd reads grad_hs, hs, pre, x (twice: row
# and transposed-plane producers read the same lines), writes d_x
BYTES_FWD_PREACT = 4 * T * (F + 2 * H)
BYTES_BWD_PREACT = 4 * T * (3 * H + 2 * F)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_steps(step, warmup, blocks, steps_per_block):
    """median over `blocks` of the event-timed duration of `steps_per_block` consecutive steps (ms per step)"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps_per_block):
            step()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / steps_per_block)
    out.sort()
    return out[len(out) // 2]


def extra_configs(dev, B):
    """The other BASELINE.json configurations on one GPU (same synthetic data shapes): each entry carries ms_per_step
    (median of 10 blocks of 20 steps enqueued back to back: a block starts on an idle queue, and with 5-step blocks that
    start was a fifth of what the fwd-only line reported), utterances/s, the algorithmic bytes of SURVEY 8(d) and the fraction of the HBM
    roof they amount to, plus fp32-equivalent FLOP/s against the fp32 MFMA peak."""
    from kws_amd import FastGRNNCUDA, RNNClassifierModel, fastgrnn_cuda
    res = {}
    g = torch.Generator().manual_seed(7)

    def entry(ms, nbytes, flops, note, paths):
        return {"ms_per_step": ms, "utt_per_s": B / (ms * 1e-3), "algorithmic_bytes_per_step": nbytes,
                "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                "tflops_f32_equiv": flops / (ms * 1e-3) / 1e12, "frac_f32_mfma_peak": flops / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                "kernel_path": paths, "workload": note}

    # config 2: dense forward only, fp32
    torch.manual_seed(0)
    m = FastGRNNCUDA(F, H, device=dev)
    x = torch.randn(T, B, F, generator=g).to(dev)
    G = torch.randn(T, B, H, generator=g).to(dev)

    def fwd_only():
        with torch.no_grad():
            m(x)
    res["fwd_only_f32"] = entry(_time_steps(fwd_only, 5, 10, 20), B * BYTES_FWD, B * FLOPS_FWD,
                                "dense fwd-only, F=32 H=128 T=99 B=%d fp32 (BASELINE config 2)" % B,
                                {"forward": fastgrnn_cuda.kernel_path(T, B, F, H, direction=0)})
    # config 3: bf16 sequences, fp32 master gradients
    xb, Gb = x.to(torch.bfloat16), G.to(torch.bfloat16)
    params = list(m.parameters())

    def step_bf16():
        for p_ in params:
            p_.grad = None
        m(xb).backward(Gb)
    res["fwd_bwd_bf16"] = entry(_time_steps(step_bf16, 5, 10, 20), B * (BYTES_FWD + BYTES_BWD) // 2, B * (FLOPS_FWD + FLOPS_BWD),
                                "dense fwd+bwd, bf16 x/hs/grad_hs/d_x, fp32 state + master gradients (BASELINE config 3)",
                                {"forward": fastgrnn_cuda.kernel_path(T, B, F, H, dtype=torch.bfloat16, direction=0, flags=4),
                                 "backward": fastgrnn_cuda.kernel_path(T, B, F, H, dtype=torch.bfloat16, direction=1, flags=4)})
    del xb, Gb
    # config 4: low-rank wRank = uRank = 16, H = 256
    HL, r = 256, 16
    torch.manual_seed(0)
    ml = FastGRNNCUDA(F, HL, wRank=r, uRank=r, device=dev)
    Gl = torch.randn(T, B, HL, generator=g).to(dev)
    pl = list(ml.parameters())

    def step_lr():
        for p_ in pl:
            p_.grad = None
        ml(x).backward(Gl)
    fl_lr = 3 * T * 2 * (F * r + r * HL + HL * r + r * HL)            # fwd 2 534 400 per utterance, x3 fwd+bwd
    res["lowrank_r16_h256"] = entry(_time_steps(step_lr, 5, 10, 20), B * 342144, B * fl_lr,
                                    "low-rank fwd+bwd, wRank=uRank=16 H=256 F=32 T=99 fp32 (BASELINE config 4)",
                                    {"forward": fastgrnn_cuda.kernel_path(T, B, F, HL, r, r, direction=0, flags=4),
                                     "backward": fastgrnn_cuda.kernel_path(T, B, F, HL, r, r, direction=1, flags=4)})
    del ml, Gl
    # the reference's default model: two dense layers 32 -> 256 -> 128, last-state Linear(128, 12), log_softmax, NLLLoss
    torch.manual_seed(0)
    C = 12
    ms = RNNClassifierModel("FastGRNNCUDA", F, 2, [256, 128], [None, None], [None, None], [1.0, 1.0], [1.0, 1.0],
                            "sigmoid", "tanh", num_classes=C, device=dev)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    pst = list(ms.parameters())

    def step_stack():
        for p_ in pst:
            p_.grad = None
        ms.init_hidden()
        ms.loss(x, y).backward()
    # the batched frame GEMM of the second layer, X[T*B,256] . W^T[256,128] -- the one place of the model where
    # `north_star` wants the matrix pipe ("MFMA used only for the batched Wx_t frame GEMM"): timed by itself through its
    # C-ABI entry, MFMA utilisation against the dense bf16 peak (every fp32 product is six bf16 MFMA terms)
    hs1 = torch.randn(T * B, 256, generator=g).to(dev)
    w2l = ms.rnn_list[1].W.detach()
    for _ in range(5):
        fastgrnn_cuda.frame_gemm(hs1, w2l)
    torch.cuda.synchronize()
    smp = []
    fastgrnn_cuda._timing = smp
    for _ in range(20):
        fastgrnn_cuda.frame_gemm(hs1, w2l)
    torch.cuda.synchronize()
    fastgrnn_cuda._timing = None
    ms_g = sorted(a.elapsed_time(b) for _, a, b in smp)[len(smp) // 2]
    fl_g = 2.0 * T * B * 256 * 128
    wx = {"kernel": "rows_gemm_split<8,8> (fastgrnn_hip_frame_gemm): P[T*B,128] = X[T*B,256] . W^T, layer 2 of the stack",
          "avg_launch_ms": ms_g, "tflops_f32_equiv": fl_g / (ms_g * 1e-3) / 1e12,
          "mfma": {"dtype": "bf16 x3 planes, 6 terms", "executed_tflops": 6 * fl_g / (ms_g * 1e-3) / 1e12,
                   "peak": PEAK_BF16_TFLOPS, "frac": 6 * fl_g / (ms_g * 1e-3) / 1e12 / PEAK_BF16_TFLOPS},
          "hbm": {"bytes": T * B * (256 + 128) * 4, "gbs": T * B * (256 + 128) * 4 / (ms_g * 1e-3) / 1e9,
                  "frac": T * B * (256 + 128) * 4 / (ms_g * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    del hs1
    fl_stack = 3 * T * 2 * (F * 256 + 256 * 256 + 256 * 128 + 128 * 128)
    res["stack_2layer"] = entry(_time_steps(step_stack, 5, 10, 20), B * (4 * T * F + 8), B * fl_stack,
                                "RNNClassifierModel 32->256->128 dense + fused head, training step (trainingConfig.py:12-15, "
                                "model.py:196-230); algorithmic bytes = read x + labels (d_x is not requested)",
                                {"layer1": [fastgrnn_cuda.kernel_path(T, B, F, 256, direction=d_, flags=4) for d_ in (0, 1)],
                                 "layer2": [fastgrnn_cuda.kernel_path(T, B, 256, 128, direction=d_, flags=4 | (256 if d_ else 0)) for d_ in (0, 1)]})
    res["stack_2layer"]["wx_gemm"] = wx
    # the same model on bf16 sequences (config 3's contract on the stack: bf16 frames and hidden-state sequences, fp32
    # state, parameters and parameter gradients)
    xb16 = x.to(torch.bfloat16)

    def step_stack_bf16():
        for p_ in pst:
            p_.grad = None
        ms.init_hidden()
        ms.loss(xb16, y).backward()
    res["stack_2layer_bf16"] = entry(_time_steps(step_stack_bf16, 5, 10, 20), B * (2 * T * F + 8), B * fl_stack,
                                     "the same model, bf16 x / hs between the layers / grad_hs, fp32 state and master gradients",
                                     {"layer1": [fastgrnn_cuda.kernel_path(T, B, F, 256, dtype=torch.bfloat16, direction=d_, flags=4) for d_ in (0, 1)],
                                      "layer2": [fastgrnn_cuda.kernel_path(T, B, 256, 128, dtype=torch.bfloat16, direction=d_, flags=4) for d_ in (0, 1)]})
    return res


def make_step(model, x, G, params, bucket, ar_events=None, sample_allreduce=lambda: False):
    """The benchmark's step: zero the gradients, module forward over [T,B,F], backward with the dense grad_hs, and -- data
    parallel -- ONE all-reduce of the flat gradient bucket (kws_amd.dp.GradBucket).  `ar_events` collects (start, end)
    events around the collective on the steps for which `sample_allreduce()` says so.  (A function of its own so that
    tests/test_dp_gloo.py can run exactly this on CPU over gloo.)"""
    def step():
        for p in params:
            p.grad = None
        hs = model(x)
        hs.backward(G)                            # L = sum(hs*G): dL/dhs = G
        if bucket is not None:
            if ar_events is not None and sample_allreduce():
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                bucket.all_reduce_()
                e1.record()
                ar_events.append((e0, e1))
            else:
                bucket.all_reduce_()
    return step


def agree_on_count(n, world, device):
    """Every step holds a collective: all ranks must run the SAME number of untimed spin-up steps (each sizes its own from
    its own clock) -- the maximum over the ranks."""
    if world <= 1:
        return int(n)
    ns = torch.tensor([int(n)], dtype=torch.int64, device=device)
    dist.all_reduce(ns, op=dist.ReduceOp.MAX)
    return int(ns.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed device spin-up before the W warmup steps (same step, every rank): a fresh GPU is "
                         "still raising its clock during the first tens of milliseconds of load")
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="N=1 only: capture the step in a HIP graph (kws_amd.GraphedStep) and time replays -- the same "
                         "kernels with ~10 us of host work per step instead of 130-270 us; reported as host: hip_graph")
    ap.add_argument("--io", choices=["f32", "bf16"], default="f32",
                    help="sequence dtype: f32 (default; the reference's type) or bf16 frames/hidden states/grad_hs "
                         "with fp32 state, parameters and parameter gradients (BASELINE config 'bf16 with fp32 master grads')")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the other BASELINE configurations (fwd-only, bf16, low-rank, 2-layer stack) at N=1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from kws_amd import FastGRNNCUDA, fastgrnn_cuda
    from kws_amd.dp import GradBucket

    B = args.batch
    torch.manual_seed(0)                         # identical parameters on every rank
    model = FastGRNNCUDA(F, H, device=dev)       # reference init: 0.1*randn, biases 1, zeta 1, nu -4
    g = torch.Generator().manual_seed(1000 + rank)
    seq_dtype = torch.bfloat16 if args.io == "bf16" else torch.float32
    x = torch.randn(T, B, F, generator=g).to(seq_dtype).to(dev)
    G = torch.randn(T, B, H, generator=g).to(seq_dtype).to(dev)
    params = [p for p in model.parameters()]
    bucket = GradBucket(params, world) if world > 1 else None

    ar_events = []                                # (start, end) around the gradient all-reduce, every fourth step
    step = make_step(model, x, G, params, bucket, ar_events, lambda: fastgrnn_cuda._timing is not None)

    eager_step = step
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is an N=1 option (the gradient all-reduce stays outside the captured step)")
        from kws_amd import GraphedStep
        step = GraphedStep(eager_step)

    # Spin-up (untimed, reported as spinup_ms): a GPU that has been idle needs tens of milliseconds of load before its
    # clock settles -- with 5-10 warmup steps (2-4 ms) the timed region of a fresh box measured the ramp (0.41-0.47
    # ms per step, backward launches of 270-300 us) instead of the kernels (0.37 ms, 235 us; same process, later
    # blocks).  The contract's W warmup steps and K timed steps follow unchanged.
    if args.spinup_ms > 0:
        for _ in range(3):                                 # first calls: library load, plan and workspace caches
            step()
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        est_ms = max(1e-3, (time.perf_counter() - t_spin) * 1e3 / 10)
        n_spin = agree_on_count(int(args.spinup_ms / est_ms), world, dev)   # (every step holds a collective)
        for _ in range(n_spin):                            # enqueued back to back: continuous load, no host gaps
            step()
    for _ in range(args.warmup):
        step()
    # HIP events around the operator's launches (on the launch stream) for the roofline's kernel durations: on every
    # FOURTH step of the timed region only -- a timing event is a marker the queue has to retire before the next
    # dispatch, and with one around every launch the step itself ran 10 % slower (0.417 vs 0.373 ms) than the
    # same loop without them
    samples = []
    SAMPLE_EVERY = int(os.environ.get("BENCH_SAMPLE_EVERY", "4"))
    # wait for the spin-up / warmup queue by POLLING the stream (the host thread stays awake and returns the moment the
    # GPU is done: a blocking synchronise sleeps, and the milliseconds until it is scheduled again are idle time the
    # clock governor answers -- DESIGN.md 5), then the contract's synchronise + barrier + synchronise
    _st = torch.cuda.current_stream(dev)
    while not _st.query():
        pass
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fastgrnn_cuda._timing = samples if (i % SAMPLE_EVERY == 0 or args.steps < 2 * SAMPLE_EVERY) else None
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing, fastgrnn_cuda._timing = samples, None
    if args.graph:                  # events cannot be recorded inside a replay: sample the kernel durations on eager steps
        torch.cuda.synchronize()
        for i in range(8):
            fastgrnn_cuda._timing = samples
            eager_step()
        torch.cuda.synchronize()
        timing, fastgrnn_cuda._timing = samples, None
    # (beside the contract's single timed region: the median over 10 further blocks of the same K steps, so that a
    # 10-20 ms sample is not the only number -- reported as ms_per_step_median_of_blocks, never as `value`)
    blocks = []
    for _ in range(10):
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - tb) / args.steps)
    blocks.sort()
    ms_median_blocks = 1e3 * blocks[len(blocks) // 2]
    # second headline leg (N = 1, default run): the SAME step captured once in a HIP graph (kws_amd.GraphedStep) and
    # replayed -- the same kernels with ~10 us of host work per step instead of the eager path's 120-270, i.e. what the
    # GPU side alone allows.  Same protocol: W warmup replays, K timed replays between synchronises, and the median
    # of ten further K-step blocks.
    graph_leg = None
    if world == 1 and not args.graph:
        try:
            from kws_amd import GraphedStep
            gstep = GraphedStep(eager_step)
            # (the capture leaves the queue idle for a while: the same untimed spin-up as the eager leg, then W warmups)
            for _ in range(int(args.spinup_ms / max(1e-3, ms_median_blocks)) + max(args.warmup, 5)):
                gstep()
            _st = torch.cuda.current_stream(dev)
            while not _st.query():
                pass
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(args.steps):
                gstep()
            torch.cuda.synchronize()
            dtg = time.perf_counter() - tg
            gblocks = []
            for _ in range(10):
                torch.cuda.synchronize()
                tb = time.perf_counter()
                for _ in range(args.steps):
                    gstep()
                torch.cuda.synchronize()
                gblocks.append((time.perf_counter() - tb) / args.steps)
            gblocks.sort()
            graph_leg = {"host": "hip_graph_replay", "value": B * args.steps / dtg, "unit": "utterances/s",
                         "ms_per_step": 1e3 * dtg / args.steps, "ms_per_step_median_of_blocks": 1e3 * gblocks[5],
                         "steps": args.steps, "note": "kws_amd.GraphedStep: module forward + autograd backward captured once, "
                                                      "replayed; results bit-equal to the eager step (tests/test_hip_graph.py)"}
            del gstep
        except Exception as e:                     # the eager line stands on its own
            graph_leg = {"host": "hip_graph_replay", "error": repr(e)}
    ar_us = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank duration of the gradient all-reduce (events on the compute stream around bucket.all_reduce_(),
        # sampled steps of the timed region): gathered so that the first multi-GPU run explains itself
        mine = sorted(a.elapsed_time(b) * 1e3 for a, b in ar_events)
        med = mine[len(mine) // 2] if mine else float("nan")
        allr = [None] * world
        dist.all_gather_object(allr, med)
        ar_us = allr

    if rank == 0:
        ms_f = [a.elapsed_time(b) for tag, a, b in timing if tag == "forward"]
        ms_b = [a.elapsed_time(b) for tag, a, b in timing if tag == "backward"]
        avg_f = sum(ms_f) / len(ms_f)
        avg_b = sum(ms_b) / len(ms_b)
        value = world * B * args.steps / dt
        path_f = fastgrnn_cuda.kernel_path(T, B, F, H, direction=0)
        path_b = fastgrnn_cuda.kernel_path(T, B, F, H, direction=1)
        tf_b = B * FLOPS_BWD / (avg_b * 1e-3) / 1e12
        tf_f = B * FLOPS_FWD / (avg_f * 1e-3) / 1e12
        split = path_b == 2
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("backward_unroll_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "utterances/sec fwd+bwd, T=99 feat=32 hidden=128, bs=4096 at 1/2/4/8 GPUs",
            "value": value, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "spinup_ms": args.spinup_ms,
            "ms_per_step": 1e3 * dt / args.steps, "ms_per_step_median_of_blocks": ms_median_blocks,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 sequences (x, hs, grad_hs, d_x); fp32 state, parameters, gradients (split-precision MFMA, fp32 accumulate)"
                      if args.io == "bf16" else
                      "f32 (split-precision MFMA: exact 3xbf16 planes, fp16 two-plane forward state product; fp32 accumulate)" if split else "f32"),
            "data": "synthetic", "host": "hip_graph_replay" if args.graph else "eager",
            "graph_replay": graph_leg,
            "allreduce_us": ({"per_rank_median": ar_us, "bytes": 4 * sum(bucket.sizes),
                              "note": "events around GradBucket.all_reduce_() on the compute stream (one flat fp32 bucket, "
                                      "ReduceOp.AVG on RCCL)"} if ar_us is not None else None),
            "config": {"workload": "FastGRNN dense fwd+bwd training step (FastGRNNCUDA module + autograd), T=99 F=32 "
                                   "H=128 B=%d per GPU, fp32 results, dense grad_hs; %s" % (
                                       B, "split-precision MFMA kernels, one saved [T,B,H] tensor" if split
                                       else "z_s/h_prime_s saved as the reference operator does"),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "kernel_path": {"forward": path_f, "backward": path_b}},
            # Dominant kernel = the backward scan (fastgrnn_hip_backward_unroll).  SURVEY 8d designates HBM
            # bandwidth as the roof of the scan: achieved = algorithmic bytes per launch (126 720 B per
            # utterance: read grad_hs, hs, x; write d_x) / average launch time measured with events on the launch
            # stream inside the timed region; traffic = the PMC byte count of profiles/traffic.json; moved_gbs
            # prices the bytes the one-saved-tensor autograd path really moves.  The kernel is NOT HBM-bound:
            # roofline_mfma gives the matrix-pipe view -- algorithmic fp32 FLOP against the fp32 MFMA (= fp32
            # vector) peak, and "executed": every fp32 product is issued as 6 bf16 MFMA terms, priced against the
            # dense bf16 peak.  What binds is the SIMD's VALU-type issue slot.
            "roofline": {"kernel": "backward_unroll", "bound": "hbm",
                         "achieved": B * BYTES_BWD / (avg_b * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": B * BYTES_BWD / (avg_b * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                         "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "workload (tools/profile.sh; counters cannot be read from inside the run)",
                         "avg_launch_ms": avg_b, "bytes_per_launch": B * BYTES_BWD,
                         "moved_gbs": (B * BYTES_BWD_PREACT / (avg_b * 1e-3) / 1e9) if split else None},
            "roofline_mfma": {"kernel": "backward_unroll", "bound": "mfma", "achieved": tf_b, "peak": PEAK_F32_TFLOPS,
                              "unit": "TFLOP/s", "frac": tf_b / PEAK_F32_TFLOPS, "flops_per_launch": B * FLOPS_BWD,
                              "executed": ({"dtype": "bf16 x3 planes, 6 terms", "tflops": 6 * tf_b,
                                            "peak": PEAK_BF16_TFLOPS, "frac": 6 * tf_b / PEAK_BF16_TFLOPS} if split else None)},
            "forward_kernel": {"avg_launch_ms": avg_f, "tflops": tf_f, "frac_f32_peak": tf_f / PEAK_F32_TFLOPS,
                               "hbm_gbs": B * BYTES_FWD / (avg_f * 1e-3) / 1e9},
        }
        if world == 1 and not args.no_extra_configs:
            print("[bench] headline done: %.0f utt/s; timing the other BASELINE configurations ..." % value,
                  file=sys.stderr, flush=True)
            del x, G
            out["configs"] = extra_configs(dev, B)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.fastgrnn_torch_port import time_fwd_bwd
            # the GPU box gives one GPU's job a 16-core CPU share; more threads than that
            # only oversubscribes
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(16, cores))
            print("[bench] GPU leg done: %.0f utt/s; timing CPU baseline on %d threads ..." % (value, cores),
                  file=sys.stderr, flush=True)
            r = time_fwd_bwd(B, T, F, H, threads=cores, budget_s=args.cpu_budget)
            # also the README's documented CPU run (B = 64, BASELINE config 1) and a single thread
            r64 = time_fwd_bwd(64, T, F, H, threads=cores, budget_s=args.cpu_budget / 3)
            r1 = time_fwd_bwd(64, T, F, H, threads=1, budget_s=args.cpu_budget / 3)
            out["cpu_baseline"] = {"value": r["utt_per_s"], "unit": "utterances/s", "cores": r["threads"],
                                   "kind": "port", "cpu_model": _cpu_model(),
                                   "sample": "%d iterations of the full B=%d T=99 fwd+bwd step (torch CPU port of "
                                             "FastGRNNCell + BaseRNN loop + autograd), median" % (r["iters"], B),
                                   "fwd_only_value": r["fwd_utt_per_s"],
                                   "b64": {"value": r64["utt_per_s"], "cores": r64["threads"], "iters": r64["iters"],
                                           "sample": "B=64 (BASELINE config 1, README.md:54), same step"},
                                   "b64_one_thread": {"value": r1["utt_per_s"], "cores": 1, "iters": r1["iters"]}}
            out["speedup_vs_cpu"] = value / r["utt_per_s"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
