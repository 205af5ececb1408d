"""bench.py -- training samples/s (labelled + unlabelled) of the DVM-STiL training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = zero_grad -> STiLModel.training_step -> backward -> (RCCL grad all-reduce) -> Adam on one batch of
synthetic input already resident in HBM (224x224 images U[0,1), 16 categorical + 48
continuous columns, K = 286, B_l = B/8, current_epoch > start_epoch so every loss term is live, MI-layer
dropout active).  Weak scaling: every rank runs `--batch` (default 256, BASELINE.json configs[1]) samples.

`--gpus N` with N > 1 and no torch.distributed.run environment (WORLD_SIZE unset) makes THIS process a launcher: before
it touches the GPU it starts N fresh rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
rendezvous on 127.0.0.1, one GPU each), forwards rank 0's line and exits with the first failing rank's code -- so
`python bench.py --gpus 8` and the torch.distributed.run form above measure the same N-rank job (trainers/evaluate.py:170-179:
the reference leaves the launch to Lightning's DDP strategy).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the gemm_nt tile variant with the largest time share; fp32-exact MFMA
implicit GEMM): algorithmic FLOPs of its launches / their HIP-event durations, taken on the launch stream during
extra steps that run AFTER the timed region (the timed steps carry no instrumentation).  `cpu_baseline` is the oracle (a CPU port validated against the
reference) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOPS_PER_SAMPLE = {(224, 64): 41.475e9, (128, 17): 13.155e9}  # student fwd + teacher fwd + bwd
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def kernel_name(cfg: int) -> str:
    """stil_gemm_nt_config code -> the template instantiation rocprofv3 lists."""
    variant, bkd, acc2, vec, plain, bna = cfg % 100, (cfg // 100) % 10, (cfg // 1000) % 10, (cfg // 10000) % 10, (cfg // 100000) % 10, (cfg // 1000000) % 10
    b3 = (cfg // 10000000) % 10      # the opt-in split-precision (bf16x3) instantiation
    bk32, nb = bkd >= 1, (1 if bkd == 2 else 2)   # bkd: 0 = 16-deep k-tiles, 1 = 32-deep in two LDS buffers, 2 = 32-deep in one
    if variant == 44:                # the lean 128x128 kernel of the plain products
        return f"gemm_nt_big_kernel<{'true' if acc2 else 'false'}, {'true' if bna else 'false'}>"
    tm, tn = {22: (2, 2), 21: (2, 1), 12: (1, 2), 11: (1, 1)}[variant]
    b = lambda v: "true" if v else "false"  # noqa: E731
    return f"gemm_nt_kernel<{tm}, {tn}, {32 if bk32 else 16}, {b(vec)}, {b(acc2)}, {b(plain)}, {b(bna)}, {nb}, {b(b3)}>"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (labelled + unlabelled)")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--ncat", type=int, default=16)
    ap.add_argument("--ncon", type=int, default=48)
    ap.add_argument("--classes", type=int, default=286)
    ap.add_argument("--variant", choices=["dvm", "saint", "cardiac", "mmatch", "comatch", "simmatch", "freematch"], default="dvm",
                    help="dvm = BASELINE configs[1..2] (the bench line); saint = config 4; cardiac = config 5 (26 cat + 49 con, K=2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU baseline (BASELINE.md section 3: 32; 256 = the bench's own batch, ~2 min per step)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps after one warm-up (BASELINE.md section 3: >= 3)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (driver.GraphedTrainStep)")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="fp32",
                    help="fp32 = fp32-exact MFMA everywhere (the bench line, `value`); bf16x3 = the OPT-IN split-precision mode of the NT "
                         "products (three bf16 terms per operand, six bf16 MFMAs, fp32 accumulate: at or below the fp32 chain's distance from "
                         "float64); its line carries `precision` and is never the headline")
    ap.add_argument("--launch", choices=["auto", "eager", "graph"], default="auto",
                    help="auto: hipGraph replay when the per-GPU step is launch-bound (driver.wants_graph: batch x pixels), eager otherwise; "
                         "--graph = --launch graph")
    ap.add_argument("--host-input", action="store_true", help="every step's batch starts in pageable host memory and goes through "
                    "data.DevicePrefetcher (PCIe-inclusive rate; NOT the contract's `value`, which has inputs resident in HBM)")
    ap.add_argument("--breakdown", action="store_true", help="also print per-entry-point GPU time of the last step (stderr)")
    ap.add_argument("--cpu-b256", action="store_true", help="also time the CPU baseline at the bench's own batch 256 (BASELINE.md section 3; "
                    "~2 min per step) and report it as cpu_baseline_b256")
    ap.add_argument("--pipeline", choices=["resident", "device"], default="resident",
                    help="device: additionally measure the device input pipeline (augment.ContrastiveBatchBuilder on a uint8 shard resident "
                         "in HBM: crop / flip / jitter / gray / blur + tabular corruption) alone and inside the training loop; printed as a "
                         "SEPARATE JSON line before the contract line, never `value`")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="deadline (s) of the N-rank launch and of every collective's "
                    "process-group timeout: ranks still running then are terminated, their stderr tails forwarded, exit code 124")
    ap.add_argument("--launch-fault", default="", help="TEST ONLY, with --launch-check: 'raise:R' makes rank R raise after the rendezvous, "
                    "'sleep:R' makes rank R sleep past any deadline")
    ap.add_argument("--launch-check", action="store_true", help="exercise only the N-rank launch, rendezvous, barrier and max-over-ranks "
                    "timing protocol with an EMPTY step (no GPU, no model): the CPU test of the launcher; value is null")
    return ap.parse_args()


def pipeline_measure(a, m, opt, dev, fl, train_step, rank, world):
    """`--pipeline device`: SURVEY.md 8f rank 3 measured.  A uint8 HWC shard resident in HBM (what the reference's workers read
    from `.npy`, datasets/ContrastiveImagingAndTabularDataset.py:177-198) goes through augment.ContrastiveBatchBuilder per step:
    RandomResizedCrop / flip / ColorJitter / ToGray / GaussianBlur + the unaugmented view + marginal tabular corruption
    (utils/utils.py:46-70 torchvision branch -- unpinned: neither torchvision nor albumentations exists offline; :146-158
    `corrupt`, pinned bit for bit).  Returns the dict of the separate JSON line; never the contract's `value`."""
    from stil_tta_amd.augment import ContrastiveBatchBuilder
    N, P, B = 2048, a.img, a.batch
    g = torch.Generator().manual_seed(99 + rank)
    images = torch.randint(0, 256, (N, P, P, 3), dtype=torch.uint8, generator=g)
    cat = [int(c) for c in fl if int(c) != 1]
    table = torch.cat([torch.randint(0, c, (N, 1), generator=g).float() for c in cat] + [torch.randn(N, len(fl) - len(cat), generator=g)], dim=1)
    labels = torch.randint(0, a.classes, (N,), generator=g)
    target = "CAD" if a.variant == "cardiac" else "dvm"
    lab = ContrastiveBatchBuilder(images, table, labels, P, target=target, labelled=True, device=dev, seed=2022 + 1000 * rank)
    unl = ContrastiveBatchBuilder(images, table, labels, P, target=target, labelled=False, device=dev, seed=2122 + 1000 * rank)
    Bl = max(B // 8, 1)

    def build():
        return {"l": lab(torch.randint(0, N, (Bl,), generator=g)), "u": unl(torch.randint(0, N, (B - Bl,), generator=g))}

    for _ in range(2):
        build()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        build()
    torch.cuda.synchronize()
    t_alone = (time.perf_counter() - t0) / a.steps
    for _ in range(max(1, a.warmup)):
        train_step(m, opt, build())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        train_step(m, opt, build())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_loop = float(t.item()) / a.steps
    # algorithmic bytes of one batch: the uint8 source rows are gathered (read + write), read by the blur / crop kernels and
    # by the unaugmented resize; two float [B,3,P,P] outputs are written; the table is negligible
    src_b, out_b = B * P * P * 3, B * 3 * P * P * 4
    alg = 4 * src_b + 2 * out_b
    return dict(metric="device input pipeline (ContrastiveBatchBuilder: crop/flip/jitter/gray/blur + tabular corruption), batches of the bench shape",
                pipeline_ms_per_batch=round(t_alone * 1e3, 3), pipeline_samples_per_s=round(B / t_alone, 1),
                algorithmic_bytes_per_batch=alg, hbm_gbps=round(alg / t_alone / 1e9, 1),
                train_with_pipeline_samples_per_s=round(B * world / t_loop, 2), train_with_pipeline_ms_per_step=round(t_loop * 1e3, 3),
                n_gpus=world, batch_per_gpu=B, shard=f"{N} uint8 {P}x{P}x3 images + {len(fl)} columns per rank, resident in HBM",
                note="image transforms follow torchvision's branch of utils/utils.py and are UNPINNED (torchvision / albumentations / cv2 are "
                     "absent offline); tabular corruption is pinned bit for bit (tests/golden/tab_corrupt.npz); this line is never `value`")


def launch_ranks(a) -> int:
    """`--gpus N` outside torch.distributed.run: N fresh rank processes, started BEFORE this process has made any GPU call
    (a process that has initialised the GPU is never re-executed or forked).  Rank r gets LOCAL_RANK = r -> GPU r.
    The launch has a deadline (`--launch-timeout`, default 1500 s): ranks still running then are terminated (they are
    fresh children, never an exec of a GPU process), every rank's stderr tail is forwarded with the named cause, exit code 124.
    A rank that exits non-zero ends the job at once with its code (the others would wait for it in a collective)."""
    import socket
    import subprocess
    import tempfile
    ndev = int(os.environ.get("STIL_FAKE_DEVICE_COUNT") or torch.cuda.device_count())   # counting devices does not initialise the GPU
    launcher_check_devices(a.gpus, os.environ.get("STIL_DIST_BACKEND") or ("nccl" if ndev > 0 else "gloo"), ndev)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, logs = [], []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
        from stil_tta_amd.driver import host_cpu_share
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cpu_share() // a.gpus)))   # as torch.distributed.run: no N-fold host oversubscription
        log = tempfile.TemporaryFile(mode="w+")              # each rank's stderr: forwarded as a tail when the job fails
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))   # rank 0 prints the line
    rc, cause = 0, None
    deadline = time.monotonic() + a.launch_timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is None:
                    continue
                pending.remove(p)
                if c != 0 and rc == 0:
                    rc, cause = c, f"rank {procs.index(p)} exited with code {c}"
                    for q in pending:      # one rank failed: the others would wait in a collective forever
                        q.terminate()
            if pending and rc == 0 and time.monotonic() > deadline:
                rc = 124
                cause = (f"launch deadline of {a.launch_timeout:.0f} s passed with rank(s) "
                         f"{[procs.index(q) for q in pending]} still running: terminated")
                for q in pending:
                    q.terminate()
            time.sleep(0.05)
    finally:
        t_kill = time.monotonic() + 10
        for p in procs:
            while p.poll() is None and time.monotonic() < t_kill:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
    for r, log in enumerate(logs):
        log.seek(0)
        txt = log.read()
        log.close()
        if rc != 0:
            print(f"---- rank {r} (exit {procs[r].poll()}) stderr tail ----\n{txt[-2500:]}", file=sys.stderr)
        elif r == 0 and txt:
            sys.stderr.write(txt)       # a healthy job: rank 0's notes only
    if rc != 0:
        print(f"bench.py launcher: FAILED -- {cause}", file=sys.stderr, flush=True)
    return rc


def check_devices(world: int, backend: str, device_count: int, local=None, env=None) -> bool:
    """One GPU per rank.  Under RCCL ("nccl") a rank that would share its device is an error named here, before the rendezvous
    (RCCL would refuse two ranks on one device much later, with an opaque message); which launch styles are accepted is
    driver.pick_device's rule (every GPU visible to every rank, or one visible GPU per rank; LOCAL_WORLD_SIZE only when the
    launcher exports it, never WORLD_SIZE).  Under gloo ranks MAY share a device (the one-GPU test box): returns True then, and
    the bench prints `roofline: null` -- HIP events around a launch then also span the other process's time slices."""
    from stil_tta_amd.driver import pick_device
    env = os.environ if env is None else env
    local = int(env.get("LOCAL_RANK", "0")) if local is None else local
    if backend == "nccl":
        try:
            pick_device(backend, local, device_count, env)
        except RuntimeError as e:
            raise SystemExit(f"bench.py: --gpus {world}: {e}")
        return False
    lws = env.get("LOCAL_WORLD_SIZE")
    return device_count < (int(lws) if lws else world)


def launcher_check_devices(n: int, backend: str, device_count: int) -> None:
    """bench.py --gpus N as its own launcher: every rank inherits THIS process's device visibility, so N ranks under RCCL need N
    visible GPUs here (a per-rank mask cannot exist: the launcher sets none)."""
    if backend == "nccl" and device_count < n:
        raise SystemExit(f"bench.py: --gpus {n} needs {n} visible GPUs for the RCCL backend, found {device_count} "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); one process per GPU, ranks never share a device")


def launch_check(a):
    """The bench protocol with an empty step: rendezvous (gloo), barrier, K timed no-op steps, barrier, MAX over ranks, rank 0
    prints the line.  No GPU and no model: what it checks is that `--gpus N` really yields an N-rank job."""
    from stil_tta_amd.driver import init_distributed
    if os.environ.get("STIL_FAKE_DEVICE_COUNT"):     # TEST ONLY: the "ranks would share a device" refusal without a GPU
        check_devices(int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("STIL_DIST_BACKEND") or "nccl", int(os.environ["STIL_FAKE_DEVICE_COUNT"]))
    rank, world, local = init_distributed(backend="gloo", timeout_s=a.launch_timeout)
    kind, _, who = a.launch_fault.partition(":")
    if kind and int(who or -1) == rank:
        if kind == "raise":
            raise RuntimeError(f"injected failure on rank {rank} (--launch-fault)")
        if kind == "sleep":
            time.sleep(3600)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pass
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = gather_times(t, world)
    if rank == 0:
        print(json.dumps(dict(metric="training samples/sec (labeled+unlabeled) for DVM STiL", value=None, unit="samples/s", n_gpus=world,
                              steps=a.steps, warmup=a.warmup, launch_check=True, local_ranks=sorted({local}),
                              ms_per_step_by_rank=[round(x / max(1, a.steps) * 1e3, 4) for x in per_rank],
                              config=dict(global_batch=a.batch * world, parallelism=f"dp{world}"))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def gather_times(t, world):
    """Every rank's time of the timed region (the contract's ms_per_step is their MAX; the list makes a straggler visible)."""
    if world == 1:
        return [float(t.item())]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def cpu_baseline(field_lengths, classes, img, batch, steps=3):
    """The oracle's full step (training_step + backward + Adam) on the host cores: a bounded sample of the workload."""
    from oracle import stil_oracle as O
    from stil_tta_amd.driver import host_cpu_share
    torch.set_num_threads(host_cpu_share())     # this job's CPU share (cgroup quota), not the node's logical CPU count: 4.4x faster on the GPU box
    hp = O.default_hparams(field_lengths=field_lengths, num_classes=classes, img_size=img, batch_size=batch, start_epoch=0)
    sd = O.init_state(hp, seed=0)
    g = torch.Generator().manual_seed(1)
    sd["prototypes"] = torch.nn.functional.normalize(torch.randn(classes, hp.projection_dim, generator=g))
    b = O.synthetic_batch(hp, batch, seed=2022)
    opt = {}
    O.full_step(sd, opt, 1, b, hp, 1)  # warm-up
    t0 = time.perf_counter()
    for i in range(steps):
        O.full_step(sd, opt, 2 + i, b, hp, 1)
    dt = (time.perf_counter() - t0) / steps
    return dict(value=round(batch / dt, 3), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{steps} timed steps (1 warm-up) of batch {batch} at the workload's shapes "
                       f"({img}px, {len(field_lengths)} columns, K={classes}); oracle/stil_oracle.py full_step, torch {torch.__version__} CPU, "
                       f"threads = the job's CPU share (cgroup quota) of the box's {os.cpu_count()} logical CPUs",
                s_per_step=round(dt, 3))


def main():
    a = parse()
    if a.precision == "bf16x3":      # before anything imports the operator layer (and inherited by the rank processes)
        os.environ["STIL_PRECISION"] = "bf16x3"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))          # nothing below runs in the launcher; it has made no GPU call
    if a.launch_check:
        return launch_check(a)
    from stil_tta_amd import STiLModel
    from stil_tta_amd._lib import lib
    from stil_tta_amd.driver import init_distributed, train_step, synthetic_batch
    from stil_tta_amd.flat import StilAdam

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("STIL_DIST_BACKEND") or "nccl"
    ndev = torch.cuda.device_count()      # counting devices does not initialise the GPU
    assert ndev > 0 and torch.cuda.is_available(), "bench.py needs a GPU (no CPU path in the product)"
    shared_device = check_devices(world_env, backend, ndev) if world_env > 1 else False
    rank, world, local = init_distributed(timeout_s=a.launch_timeout)
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: following the launcher's environment", file=sys.stderr)
    local = local % ndev if shared_device else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    extra = {}
    if a.variant == "saint":
        extra = dict(tabular_encoder="saint")
    elif a.variant == "cardiac":  # configs/config_cardiac_STiL.yaml deltas
        a.ncat, a.ncon, a.classes = 26, 49, 2
        extra = dict(target="CAD", th1=0.85, beta=1.0, gamma=1.0, rate_pseudo=0.95, ema_momentum=0.4, lr_eval=1e-3)
    fl = [(4 if a.variant == "cardiac" else 8)] * a.ncat + [1] * a.ncon
    torch.manual_seed(2022)
    if a.variant == "mmatch":  # the MMatch baseline of the reference on the same kernels (configs/config_dvm_MMatch.yaml)
        from stil_tta_amd import MMatch as STiLModel  # noqa: F811
        extra = dict(DA=True, alpha=1.0, mmatch_lambda=5.0)
    match = a.variant in ("comatch", "simmatch", "freematch")  # the Match baselines (configs/config_dvm_Multi*Match.yaml) on the same kernels
    if match:
        import stil_tta_amd
        STiLModel = {"comatch": stil_tta_amd.CoMatch, "simmatch": stil_tta_amd.SimMatch, "freematch": stil_tta_amd.FreeMatch}[a.variant]  # noqa: F811
        extra = dict(K=2560, DA=True)
    m = STiLModel(dict(field_lengths=fl, num_classes=a.classes, img_size=a.img, batch_size=a.batch, start_epoch=35,
                       repeat_ratio=1.0, seed=2022 + rank, **extra))
    m.setup_device(dev)
    m.train()
    m.current_epoch = 36  # > start_epoch: every loss term of STiLModel.py:345 is live
    g = torch.Generator().manual_seed(7)
    if a.variant != "mmatch" and not match:
        m.prototypes.copy_(torch.nn.functional.normalize(torch.randn(a.classes, 128, generator=g)).to(dev))
    opt = StilAdam(m.flat, lr=1e-4)
    batch = synthetic_batch(fl, a.classes, a.batch, a.img, seed=2022 + rank, device=dev)
    if match:  # (x, y, index) / ((weak, strong[, strong2]), y) of trainers/evaluate.py:50-83, from the same synthetic tensors
        (_, iml), (_, tabl), yl = batch["l"][0], batch["l"][1], batch["l"][2]
        (_, imu), (_, tabu), yu = batch["u"][0], batch["u"][1], batch["u"][2]
        views = [(imu, tabu), (imu.flip(3).contiguous(), tabu)] + ([(imu.flip(2).contiguous(), tabu)] if a.variant == "comatch" else [])
        batch = {"l": ((iml, tabl), yl, torch.arange(len(yl), device=dev)), "u": (views, yu)}

    from stil_tta_amd.driver import wants_graph
    a.graph = a.graph or a.launch == "graph" or (a.launch == "auto" and world == 1 and a.variant in ("dvm", "saint", "cardiac") and not a.host_input
                                                 and a.pipeline == "resident" and wants_graph(a.batch, a.img))
    if a.graph:
        from stil_tta_amd.driver import GraphedTrainStep
        gstep = GraphedTrainStep(m, opt, batch, warmup=max(1, a.warmup))
        eager_step = train_step
        train_step = lambda m_, o_, b_: gstep(b_)  # noqa: E731
    feed = None
    if a.host_input:
        from stil_tta_amd.data import DevicePrefetcher
        host_batch = synthetic_batch(fl, a.classes, a.batch, a.img, seed=2022 + rank, device="cpu")
        feed = iter(DevicePrefetcher((host_batch for _ in range(a.warmup + a.steps)), dev, depth=2))
    for _ in range(a.warmup):
        train_step(m, opt, next(feed) if feed else batch)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L = lib()
    prof = prof_conc = None
    for s in range(a.steps):      # the timed region carries no instrumentation
        train_step(m, opt, next(feed) if feed else batch)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = gather_times(torch.tensor([dt], dtype=torch.float64, device=dev), world)
    dt = max(per_rank)            # the contract: MAX over ranks
    # OUTSIDE the timed region, two more steps.  (1) the step as timed (both streams), HIP events around the GEMM launches on
    # whichever stream they go to: the kernel while the side stream co-runs (`two_stream`).  (2) every launch kept on ONE
    # stream and bracketed by HIP events: clean per-kernel durations (no second kernel sharing the chip), which is what
    # the roofline of the kernel is priced on.
    # Ranks that share a device (gloo on the one-GPU test box) skip both: an event pair then also spans the OTHER process's
    # time slices, so it prices nothing (`roofline: null` with the reason).
    if not shared_device:
        if not a.graph:
            L.begin_profile(single_stream=False, only=("gemm_nt",))
            train_step(m, opt, batch)
            prof_conc = L.end_profile()
        L.begin_profile(only=None if a.breakdown else ("gemm_nt", "wgrad_tn"))
        (eager_step if a.graph else train_step)(m, opt, batch)
        torch.cuda.synchronize()
        prof = L.end_profile()
    loss = float(m.last["loss"].detach())
    assert loss == loss, "loss is NaN"
    if a.pipeline == "device" and a.variant in ("dvm", "saint", "cardiac"):
        pl = pipeline_measure(a, m, opt, dev, fl, eager_step if a.graph else train_step, rank, world)
        if rank == 0:
            print(json.dumps(pl), flush=True)

    if rank == 0:
        value = a.batch * world * a.steps / dt
        # dominant kernel = the gemm_nt instantiation with the largest share of the step's GPU time; meta[0] is the
        # configuration the library reports for that very launch (stil_gemm_nt_config: tile, BK, vector loads, two-level sums)
        byvar = {}
        for name, ms, meta in prof or ():
            if name == "gemm_nt" and meta:
                c = byvar.setdefault(meta[0], [0, 0.0, 0.0, 0.0])
                c[0] += 1; c[1] += ms * 1e-3; c[2] += meta[1]; c[3] += meta[3]
        roof = None
        roof_reason = ("ranks share a device (gloo backend, fewer GPUs than ranks): HIP events around a launch also span the other "
                       "process's time slices" if shared_device else "no gemm_nt launch was recorded")
        if byvar:
            var, (nl, tsum, fsum, bsum) = max(byvar.items(), key=lambda kv: kv[1][1])
            assert tsum > 0 and fsum > 0, (var, nl, tsum, fsum)
            kname = kernel_name(var)
            achieved = fsum / tsum / 1e12
            roof = dict(bound="mfma", kernel=kname, achieved=float(f"{achieved:.6g}"), peak=PEAK_FP32_MFMA_TFLOPS,
                        unit="TFLOP/s", frac=float(f"{achieved / PEAK_FP32_MFMA_TFLOPS:.6g}"), traffic=None,
                        launches_per_step=nl, avg_launch_us=round(tsum / max(1, nl) * 1e6, 2),
                        flops_per_step=fsum, share_of_step_time=round(tsum / (dt / a.steps), 4),
                        algorithmic_bytes=round(bsum / max(1, nl)),
                        all_gemm_nt={kernel_name(k): dict(launches=v[0], ms=round(v[1] * 1e3, 3), tflops=round(v[2] / v[1] / 1e12, 2)) for k, v in
                                     sorted(byvar.items(), key=lambda kv: -kv[1][1])},
                        note="achieved / avg_launch_us: HIP events around every launch of one extra step run right after the timed region "
                             "with all launches on ONE stream (agrees with profiles/*_single_stream.csv = this command with "
                             "STIL_WGRAD_STREAM=0); two_stream: the same kernel during another extra step, while the side stream (EMA "
                             "teacher, weight gradients) co-runs -- what a rocprofv3 trace of the default command averages "
                             "(profiles/*_bench_kernel_stats.csv)")
            # HBM traffic of that kernel: PMC counters cannot be read from inside the process; the latest separate-pass
            # rocprofv3 measurement of this same command is kept under profiles/ and quoted ONLY while it is for this kernel AND was
            # taken on these very kernel sources (kernel_source_sha = content hash of csrc/: the box has no .git to ask).
            try:
                from stil_tta_amd._lib import source_hash
                sha = source_hash()
                pdir = os.path.join(ROOT, "profiles")
                pmc = sorted(f for f in os.listdir(pdir) if f.endswith("pmc_traffic.json"))
                tj = json.load(open(os.path.join(pdir, pmc[-1]))) if pmc else None
                # quoted only while it describes THESE launches: same kernel, same sources (kernels + the host files that compose a
                # launch, _lib.source_hash) and -- belt and braces -- the same algorithmic bytes per launch as this run's
                alg_ok = tj is not None and (tj.get("algorithmic_bytes_per_launch") is None or
                                             abs(tj["algorithmic_bytes_per_launch"] - roof["algorithmic_bytes"]) <= 0.005 * roof["algorithmic_bytes"])
                if tj and tj["kernel"] == kname and tj.get("kernel_source_sha") == sha and alg_ok and a.batch == 256 and a.img == 224 and a.variant == "dvm":
                    roof["traffic"] = round(tj["bytes_per_launch"])
                    roof["traffic_source"] = f"profiles/{pmc[-1]} (sources {sha})"
                mu = sorted(f for f in os.listdir(pdir) if f.endswith("mfma_util.json"))
                uj = json.load(open(os.path.join(pdir, mu[-1]))) if mu else None
                if uj and kname in uj.get("kernels", {}) and uj.get("kernel_source_sha") == sha and a.batch == 256 and a.img == 224:
                    roof["mfma_busy"] = uj["kernels"][kname]["mfma_busy_of_cu_busy"]
                    roof["mfma_busy_source"] = (f"profiles/{mu[-1]} (kernel sources {sha}; SQ_VALU_MFMA_BUSY_CYCLES / 4 / SQ_BUSY_CU_CYCLES, "
                                                "rocprofv3 --pmc on this command)")
            except Exception:
                pass
            if prof_conc:  # the same kernel while the second stream co-runs (what a rocprofv3 trace of this command averages)
                cn, ct, cf = 0, 0.0, 0.0
                for name, ms, meta in prof_conc:
                    if name == "gemm_nt" and meta and meta[0] == var:
                        cn += 1; ct += ms * 1e-3; cf += meta[1]
                if cn:
                    roof["two_stream"] = dict(avg_launch_us=round(ct / cn * 1e6, 2), achieved=round(cf / ct / 1e12, 2), launches=cn)
        fps = FLOPS_PER_SAMPLE.get((a.img, a.ncat + a.ncon))
        out = dict(metric="training samples/sec (labeled+unlabeled) for DVM STiL", value=round(value, 2), unit="samples/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype=("f32" if a.precision == "fp32" else "f32 via bf16x3 split products"), data="synthetic",
                   config=dict(workload=f"config_{'cardiac' if a.variant == 'cardiac' else 'dvm'}_{dict(mmatch='MMatch', comatch='MultiCoMatch', simmatch='MultiSimMatch', freematch='MultiFreeMatch').get(a.variant, 'STiL')}{'_SAINT' if a.variant == 'saint' else ''} ResNet-50 + {'SAINT' if a.variant == 'saint' else 'Transformer'} tabular, B={a.batch}/GPU "
                                        f"({a.batch // 8} l + {a.batch - a.batch // 8} u), {a.img}px + {a.ncat + a.ncon} cols, K={a.classes}, "
                                        f"pseudo-label phase, MI dropout on",
                               global_batch=a.batch * world, parallelism=f"dp{world}", precision=("fp32-exact MFMA" if a.precision == "fp32" else
                                          "bf16x3 split-precision NT products (OPT-IN; weight gradients fp32-exact MFMA) -- not the headline"),
                               launch="hipGraph replay" if a.graph else "eager", split_k=bool(__import__("stil_tta_amd.ops", fromlist=["_SPLITK"])._SPLITK),
                               input="host memory through data.DevicePrefetcher (PCIe-inclusive)" if a.host_input else "resident in HBM"),
                   ms_per_step_by_rank=[round(x / a.steps * 1e3, 3) for x in per_rank],
                   roofline=roof, loss=round(loss, 5),
                   hbm_peak_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 2), hbm_reserved_gb=round(torch.cuda.memory_reserved(dev) / 1e9, 2))
        if roof is None:
            out["roofline_reason"] = roof_reason
        if fps:
            out["step_tflops_algorithmic"] = round(value * fps / 1e12, 2)
            out["step_frac_of_fp32_mfma_peak"] = round(value * fps / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4)
        if a.breakdown:
            agg = {}
            for name, ms, meta in prof or ():
                key = name if name != "gemm_nt" else kernel_name(meta[0])
                c = agg.setdefault(key, [0, 0.0])
                c[0] += 1; c[1] += ms
            tot = sum(v[1] for v in agg.values())
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"  {k:28s} {v[0]:5d} launches {v[1]:9.3f} ms {100 * v[1] / tot:5.1f}%", file=sys.stderr)
            print(f"  sum of C-ABI kernels {tot:.3f} ms of {dt / a.steps * 1e3:.3f} ms/step", file=sys.stderr)
            shapes = {}
            for name, ms, meta in prof or ():
                if name in ("gemm_nt", "wgrad_tn") and meta:
                    c = shapes.setdefault(meta[2], [0, 0.0, 0.0])
                    c[0] += 1; c[1] += ms; c[2] += meta[1]
            print("  gemm by shape (M, N, K, k, stride, mode; mode 2 = wgrad_tn): launches, ms, TFLOP/s", file=sys.stderr)
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:60]:
                print(f"    {str(k):44s} {v[0]:3d} {v[1]:8.3f} ms {v[2] / v[1] / 1e9:7.1f} TF", file=sys.stderr)
        if world == 1 and not a.no_cpu_baseline and a.variant == "dvm":
            out["cpu_baseline"] = cpu_baseline(fl, a.classes, a.img, a.cpu_batch, max(1, a.cpu_steps))
            if a.cpu_b256 and a.cpu_batch != 256:
                out["cpu_baseline_b256"] = cpu_baseline(fl, a.classes, a.img, 256, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
