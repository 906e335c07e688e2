"""bench.py — samples/s of a DLRM training step on Criteo-1TB-shaped synthetic input, 1..8 MI355X.

Metric (BASELINE.json): samples/sec, Criteo-1TB DLRM, global batch 65 536, at 1/2/4/8 GPUs, plus
the fraction of the HBM roofline reached by the dominant embedding kernel, plus the reference's
CPU EmbeddingBagCollection timed on this box's host cores.

One "step" = one full training pass over one global batch: TBE forward (HIP), pooled exchange
(RCCL all-to-all when N > 1), dense MLPs + dot interaction + BCE loss (fp32, rocBLAS/hipBLASLt),
backward, TBE backward with the fused exact-SGD update (HIP), dense SGD step.  Inputs are
resident in HBM when the timed region starts.  Scaling is STRONG: the global batch stays
65 536 and is split over the ranks, as in the reference's published 8-GPU run
(examples/dlrm/README.MD:38-45).

Launch: `python bench.py --gpus N --steps K --warmup W` for any N.  With N > 1 and no WORLD_SIZE in
the environment this process starts N fresh rank processes itself (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, the command of examples/dlrm/README.MD:17-28)
BEFORE anything here touches the GPU (torch is not even imported in the parent), relays rank 0's ONE JSON
line and exits with the ranks' return code.  Launched by an external torchrun (WORLD_SIZE set) it is a rank.
"""
import argparse
import contextlib
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "torchrec-oldfork_amd")  # fbgemm_gpu/ + torchrec_amd/ (importing them sets HSA_ENABLE_IPC_MODE_LEGACY=0
#                                                   before HIP initialises: RCCL needs dmabuf IPC on this platform)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the on-box copy rate is measured per run
#                           (roofline.peak_measured_copy_GBs)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32 matrix peak (MI355X_MICROARCH.md)
XGMI_LINK_GBS = 153.0     # one xGMI link, one direction; 7 links per GPU
D = 128
CRITEO_F = 26
F = CRITEO_F
# SURVEY.md §8(d) algorithmic bytes per sample (fp32 rows, int64 indices/offsets, L = 1)
BYTES_FWD = F * (D * 4 + 8) + F * 8 + F * D * 4          # 27 040
BYTES_BWD_SGD = F * D * 4 + F * 16 + 2 * F * D * 4       # 40 352
MFLOP_PER_SAMPLE_TRAIN = 14.75                           # 4.917 MFLOP forward x 3


@contextlib.contextmanager
def native_stdout_to_stderr():
    """RCCL prints a version banner on the C-level stdout when a communicator is created; the contract is ONE
    JSON line on stdout, so file descriptor 1 points at stderr while the process group comes up."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--global-batch", type=int, default=65536)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--row-cap", type=int, default=0, help="debug: cap rows per table")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf alpha for ids (0 = uniform)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0,
                    help="time budget of the CPU baseline (it runs >= 10 train iterations per batch size whatever this says)")
    ap.add_argument("--num-batches", type=int, default=32, help="distinct pre-generated batches (SURVEY.md §8d: >= 32)")
    ap.add_argument("--seed", type=int, default=0, help="dense-parameter and table initialisation (tables are initialised as a "
                    "function of (seed, table, global row): any sharding starts from the same model)")
    ap.add_argument("--data-ranks", type=int, default=0,
                    help="draw the batches as this many ranks would (rank r seeds its generator with 1234 + r, as the reference's "
                         "examples do) and give every real rank its contiguous share: --gpus 1 --data-ranks 2 trains on the "
                         "global batches of a --gpus 2 run.  0 = the number of ranks")
    ap.add_argument("--row-wise", type=int, default=-1,
                    help="shard the N largest tables row-wise.  Default (-1): at N > 1 the pinned mixed plan of BASELINE "
                         "config 3 / SURVEY.md §8d (the 4 largest tables row-wise, the rest table-wise or replicated), "
                         "at N = 1 none.  0 = the planner's own choice (row-wise only for capacity)")
    ap.add_argument("--rw-input-dist", choices=["auto", "windows", "bucketize"], default="auto",
                    help="input dist of row-wise features: row windows (sync-free, every rank receives every id) or the "
                         "reference's bucketized exchange; auto = windows at this workload's pooling factor of 1")
    ap.add_argument("--spawn", action="store_true",
                    help="start the rank processes through the built-in launcher even for --gpus 1")
    ap.add_argument("--tuned-gemms", choices=["on", "off"], default="on",
                    help="replay the recorded hipBLASLt / rocBLAS kernel choice per GEMM shape (torchrec_amd/tuning)")
    ap.add_argument("--hip-graphs", choices=["auto", "on", "off"], default="auto",
                    help="replay the collective-free dense segments from HIP graphs (auto: per-rank batch <= 32768, "
                         "where host launches show)")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args) -> int:
    """Parent of an N-rank run: starts N fresh rank processes (one per GPU) and relays rank 0's JSON line.
    Nothing in this process has imported torch or touched the GPU."""
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += [a for a in sys.argv[1:] if a != "--spawn"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(port)
    env["TORCHREC_AMD_BENCH_LAUNCHER"] = "bench.py"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr: inherited
    line = None
    for raw in proc.stdout:
        txt = raw.strip()
        if txt.startswith("{") and '"metric"' in txt:
            line = txt  # rank 0's result; anything else a rank printed goes to stderr
        else:
            sys.stderr.write(raw)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("[bench] the ranks exited 0 but printed no result line", file=sys.stderr)
        return 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def dry_run(args) -> None:
    """TORCHREC_AMD_BENCH_DRYRUN=1: the launcher / rendezvous / timing / reporting skeleton of a rank on CPU
    over gloo, no model and no GPU (tests/test_bench_launcher.py drives it at N = 2).  =fail: rank 1 exits 3."""
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("TORCHREC_AMD_BENCH_DRYRUN") == "fail" and rank == world - 1:
        sys.exit(3)
    x = torch.ones(4)
    for _ in range(args.warmup):
        dist.all_reduce(x)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dist.all_reduce(x)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "samples/sec Criteo-1TB DLRM batch 65536", "dry_run": True, "n_gpus": world,
                          "ranks": dist.get_world_size(), "backend": dist.get_backend(), "steps": args.steps,
                          "warmup": args.warmup, "value": args.global_batch * args.steps / float(t.item()),
                          "launcher": os.environ.get("TORCHREC_AMD_BENCH_LAUNCHER", "external")}), flush=True)
    dist.destroy_process_group()


def read_profile(lib, slot):
    tot, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.tbe_profile_read(slot, ctypes.byref(tot), ctypes.byref(n))
    return tot.value, n.value


def measure_copy_GBs(torch, dev) -> float:
    """On-box device copy rate: a 1-GiB device-to-device copy, bytes read + written over the best of 5 (the ceiling a
    pure streaming kernel reaches here, shown next to the 8 TB/s spec peak: SURVEY.md §8d)."""
    n = 1 << 28  # 2^28 fp32 = 1 GiB
    src = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    best = float("inf")
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src)
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    del src, dst
    return 2 * n * 4 / (best * 1e-3) / 1e9


def main(args):
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    n_ranks = int(os.environ.get("WORLD_SIZE", "1"))
    if n_ranks > 1 or os.environ.get("TORCHREC_AMD_FORCE_EXCHANGE") == "1":
        # HIP maps its streams onto this many hardware queues (default 4).  A rank of an N > 1 run uses 7 streams (compute,
        # memcpy, input dist, the collective's, two sort side streams, graph capture); measured on the one-rank rehearsal
        # at the 8-GPU per-rank batch: 1 queue 1.867 ms, 2: 1.754, 3: 1.740, 4 (default): 1.785, 8: 3.77 (!) per step
        # (profiles/r03_rehearsal_b8192_hw_queues.txt); no effect at N = 1 (8.55 vs 8.59 ms).  Read by the runtime when HIP
        # initialises, so it is set before torch touches the GPU; an explicit value in the environment wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
    import torch
    import torch.distributed as dist

    from fbgemm_gpu import _lib
    from torchrec_amd.datasets.random import CRITEO_1TB_ROWS, DEFAULT_CAT_NAMES, INT_FEATURE_COUNT, RandomRecDataset
    from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec_amd.distributed.model_parallel import DistributedModelParallel
    from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
    from torchrec_amd.distributed.types import ShardingEnv
    from torchrec_amd.models.dlrm import DLRMTrain
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.optim.keyed import CombinedOptimizer, KeyedOptimizerWrapper

    assert len(CRITEO_1TB_ROWS) == F
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    # TORCHREC_AMD_BENCH_BACKEND=gloo: rehearsal of an N-rank run on FEWER GPUs than ranks (ranks share devices round robin;
    # RCCL refuses two ranks on one device).  Every rank runs the product path — HIP kernels, HIP graphs, explicit step,
    # flat-gradient all-reduce; only the all-to-all is staged through the host (distributed/_rehearsal.py).  The line says
    # so (`backend`, `rehearsal`); its timings are not multi-GPU timings.
    backend = os.environ.get("TORCHREC_AMD_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("TORCHREC_AMD_BENCH_BACKEND must be nccl or gloo")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # TORCHREC_AMD_FORCE_EXCHANGE=1: rehearsal of the N > 1 data path on one GPU (a one-rank RCCL group)
    # (TORCHREC_AMD_FORCE_DDP=1 additionally wraps the dense modules in DistributedDataParallel over that group)
    # TORCHREC_AMD_FORCE_DP=1 replicates the tiny tables as an N > 1 plan does.
    rehearse = world == 1 and (os.environ.get("TORCHREC_AMD_FORCE_EXCHANGE") == "1"
                               or os.environ.get("TORCHREC_AMD_FORCE_DDP") == "1")
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        from torchrec_amd.distributed.comm import init_rccl_process_group

        with native_stdout_to_stderr():
            if rehearse:
                init_rccl_process_group(dev, rank=0, world_size=1)
                link_us = float(os.environ.get("TORCHREC_AMD_REHEARSAL_LINK_US", "0"))
                if link_us > 0:  # the pooled all-to-alls take this long on the GPU timeline, as over xGMI
                    from torchrec_amd.distributed._rehearsal import emulate_link_time

                    emulate_link_time(link_us)
            elif backend == "gloo":
                from torchrec_amd.distributed._rehearsal import stage_all_to_all_through_host

                dist.init_process_group("gloo")
                stage_all_to_all_through_host()
            else:
                init_rccl_process_group(dev)
            dist.barrier()  # the communicator (and its banner) exists before stdout is handed back
        env = ShardingEnv.from_process_group(dist.group.WORLD)
    else:
        env = ShardingEnv.from_local(1, 0)
    if args.global_batch % world:
        raise SystemExit("global batch must divide evenly over the ranks")
    data_ranks = args.data_ranks or world
    if data_ranks % world or args.global_batch % data_ranks:
        raise SystemExit("--data-ranks must be a multiple of the number of ranks and divide the global batch")
    tuned = False
    if args.tuned_gemms == "on" and os.environ.get("PYTORCH_TUNABLEOP_TUNING", "0") != "1":
        from torchrec_amd.tuning import enable_tuned_gemms
        tuned = enable_tuned_gemms()
    B_local = args.global_batch // world
    rows = [min(r, args.row_cap) if args.row_cap else r for r in CRITEO_1TB_ROWS]
    torch.manual_seed(args.seed)  # dense parameters: the same on every rank (rank 0's are broadcast anyway)

    # ---- model: examples/dlrm/dlrm_main.py:498-540 ------------------------------------------------
    tables = [EmbeddingBagConfig(name=f"t_{n}", embedding_dim=D, num_embeddings=rows[i], feature_names=[n])
              for i, n in enumerate(DEFAULT_CAT_NAMES)]
    ebc = EmbeddingBagCollection(tables=tables, device=torch.device("meta"))
    train_model = DLRMTrain(embedding_bag_collection=ebc, dense_in_features=INT_FEATURE_COUNT,
                            dense_arch_layer_sizes=[512, 256, 128],
                            over_arch_layer_sizes=[1024, 1024, 512, 256, 1], dense_device=dev)
    hip_graphs = args.hip_graphs == "on" or (args.hip_graphs == "auto" and B_local <= 32768)
    graphs_note = ("on" if hip_graphs else
                   ("off by request" if args.hip_graphs == "off" else
                    "off: per-rank batch > 32768, the GPU is busy 99 % of the step when eager (DESIGN.md §3c)"))
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology

    # plan: BASELINE config 3 is MIXED table-wise + row-wise; SURVEY.md §8d pins "the 4 largest tables row-wise,
    # 22 table-wise" so that runs are comparable.  That is the default at N > 1 (the planner's own choice would
    # shard row-wise only for capacity, i.e. not at all here: DESIGN.md §4).
    if args.row_wise >= 0:
        num_rw = args.row_wise
    else:
        num_rw = 4 if env.world_size > 1 else 0

    model = DistributedModelParallel(
        module=train_model, env=env, device=dev,
        sharders=[EmbeddingBagCollectionSharder(fused_params={"learning_rate": args.lr}, rw_input_dist=args.rw_input_dist)],
        planner=EmbeddingShardingPlanner(Topology(env.world_size), num_row_wise=num_rw or None),
        init_data_parallel=False)
    shard = model.sharded_modules()[0]
    shard.reset_parameters_sharding_invariant(args.seed)
    if rehearse and float(os.environ.get("TORCHREC_AMD_REHEARSAL_LINK_US", "0")) > 0 and shard._exchange:
        from torchrec_amd.distributed._rehearsal import set_link_full_bytes

        set_link_full_bytes(shard._exchange_layout(B_local)["recv_numel"] * 4)  # a half-batch message spins half as long
    if hip_graphs:
        # HIP-graph replay of the collective-free dense segments; captured before DistributedDataParallel
        # wraps the dense modules (distributed/train_pipeline.py explains why)
        # With a process group the segments' gradients travel through ONE flat buffer that is all-reduced
        # per segment, instead of DDP's per-parameter bucket copies (models/dlrm.py)
        failure = None
        try:
            train_model.capture_hip_graphs(B_local, flat_grads=True, process_group=env.process_group)
        except Exception as e:  # measured run must not die on a capture problem: run the segments eagerly
            import traceback

            traceback.print_exc(file=sys.stderr)
            failure = f"{type(e).__name__}: {str(e)[:200]}"
            print(f"[bench] HIP-graph capture failed ({failure}); running eagerly", file=sys.stderr, flush=True)
        if world > 1:
            # every rank must take the same path (graphs reduce the dense gradients through the flat buffer, the eager
            # fallback through DistributedDataParallel): one rank's failure is everybody's
            ok = torch.tensor([0 if failure else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and failure is None:
                failure = "another rank's capture failed"
        if failure is not None:
            graphs_note = f"FELL BACK to eager: capture failed with {failure}"
            train_model._graphs = None
            if hasattr(train_model, "_flat_dense"):
                object.__delattr__(train_model, "_flat_dense")
            ebc_now = train_model.model.sparse_arch.embedding_bag_collection
            if hasattr(ebc_now, "set_output_buffer"):
                ebc_now.set_output_buffer(None)
            if hasattr(ebc_now, "set_replicated_grad_sink"):
                ebc_now.set_replicated_grad_sink(None)
            torch.cuda.synchronize()
            hip_graphs = False
    model.init_data_parallel()
    dense_params = dict(model.named_parameters())
    optimizer = CombinedOptimizer([
        model.fused_optimizer,
        # torch.optim.SGD, or its one-kernel form when the parameters sit in one flat buffer (N > 1 with HIP graphs)
        KeyedOptimizerWrapper(dense_params, lambda p: train_model.dense_optimizer(p, lr=args.lr))])
    plan = model.plan
    kinds = [p.sharding_type for p in next(iter(plan.plan.values())).values()]
    n_rw, n_dp = kinds.count("row_wise"), kinds.count("data_parallel")

    per = data_ranks // world
    data = RandomRecDataset(DEFAULT_CAT_NAMES, B_local, rows, ids_per_feature=1, num_dense=INT_FEATURE_COUNT,
                            manual_seeds=[1234 + rank * per + j for j in range(per)],
                            num_generated_batches=args.num_batches, device=dev, zipf_alpha=args.zipf or None)
    it = iter(data)
    pipe = TrainPipelineSparseDist(model, optimizer, dev)
    model.train()
    lib = _lib.load()
    copy_GBs = measure_copy_GBs(torch, dev) if rank == 0 else None

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.progress(it)
    sync_all()
    lib.tbe_profile_enable(1)
    for s in range(4):
        read_profile(lib, s)
    stamps = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    stamps[0].record()
    loss_first = loss_last = None
    for k in range(args.steps):
        out = pipe.progress(it)
        stamps[k + 1].record()
        # the loss of the first and the last timed step (a 4-byte device copy each: under HIP graphs the loss lives in
        # a static buffer the next replay overwrites); read after the timed region
        if k == 0:
            loss_first = out[0].detach().clone()
        if k == args.steps - 1:
            loss_last = out[0].detach().clone()
    sync_all()
    elapsed = time.perf_counter() - t0
    lib.tbe_profile_enable(0)
    prof = [read_profile(lib, s) for s in range(4)]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    value = args.global_batch * args.steps / elapsed
    step_ms = sorted(stamps[k].elapsed_time(stamps[k + 1]) for k in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    # ---- did the timed region compute the right thing?  ("make wrong results loud") ----------------
    # (1) no kernel gave up (sort spin-waits), no id was out of range; (2) the loss of the first / last timed step and
    # a checksum of every parameter, which tests/test_bench_rehearsal_gpu.py compares with an eager, un-pipelined,
    # graph-free replay of the same batches and with a run at another world size
    if os.environ.get("TORCHREC_AMD_BENCH_INJECT_FAULT") == "1":  # test hook: what a sort give-up during the run leaves behind
        lib.tbe_debug_inject_fault_host()
    sort_giveups = _lib.fault_count()  # everything is complete (sync_all): the fault word is final
    bounds_errors = 0
    if sort_giveups == 0:
        for m in (shard._emb_module, shard._dp_module):
            if m is not None:
                bounds_errors += m.bounds_check_errors()
    losses = torch.stack([loss_first.float().reshape(()), loss_last.float().reshape(())]).to(torch.float64)
    def sums(tensors):  # [sum, sum of |x|] in float64
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        for x in tensors:
            x = x.detach()
            acc[0] += torch.sum(x, dtype=torch.float64)
            acc[1] += torch.sum(x.abs(), dtype=torch.float64) if x.numel() < (1 << 28) else sum(
                torch.sum(c.abs(), dtype=torch.float64) for c in x.view(-1).split(1 << 28))
        return acc

    if sort_giveups == 0:
        emb_sum = sums(w for _, (w, _) in shard.local_shards().items())
        dp_sum = sums(w for _, w in shard.dp_tables().items())
    else:  # the module's accessors raise KernelFaultError now (rightly): the run is reported invalid below instead
        emb_sum, dp_sum = sums([]), sums([])
    dense_sum = sums(q for n, q in model.named_parameters() if "_dp_module" not in n)
    replicas_identical = True
    if world > 1:
        dist.all_reduce(losses)  # mean over the global batch = mean of the ranks' local means
        losses /= world
        dist.all_reduce(emb_sum)  # every sharded row lives on exactly one rank
        lo, hi = torch.cat([dense_sum, dp_sum]), torch.cat([dense_sum, dp_sum])
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))  # dense parameters + replicated tables: bit-identical replicas
        bs = torch.tensor([bounds_errors, sort_giveups], dtype=torch.int64, device=dev)
        dist.all_reduce(bs)
        bounds_errors, sort_giveups = int(bs[0].item()), int(bs[1].item())
    checks = {"sort_giveups": sort_giveups, "bounds_check_errors": bounds_errors,
              "loss_first": float(losses[0].item()), "loss_last": float(losses[1].item()),
              # sum and sum of |x| over every dense parameter / every row of every table (float64)
              "param_checksum": {"dense": dense_sum.tolist(), "embedding": (emb_sum + dp_sum).tolist()},
              "dense_replicas_identical": replicas_identical}

    # ---- roofline of the dominant embedding kernel (this rank's launches) ---------------------------
    # Units per launch on this rank: every TBE launch covers the GLOBAL batch for the features this
    # rank holds (table-wise: whole features; row-wise: 1/W of a feature's rows => 1/W of its bytes).
    # At N > 1 a rank runs TWO lookups per step (the fused one over the global batch for its sharded
    # features, the dense-gradient one over its LOCAL batch for the replicated tiny tables); the
    # profile slots hold both, so everything below is per STEP: summed launch time, summed bytes.
    kind = shard._table_kind  # -2 replicated, -1 row-wise, >= 0 owning rank
    feat_units = sum((1.0 / world) if kind[i] == -1 else (1.0 if kind[i] == rank else 0.0) for i in range(F))
    feat_units += sum(1.0 for i in range(F) if kind[i] == -2) / world  # local batch = 1/W of the global batch
    # Algorithmic bytes per launch (SURVEY.md §8d general formula, L = 1, fp32 rows, int64 ids):
    #   forward : units * B * (D*4 row + 8 id + 8 offset + D*4 output)
    #   backward: units * B * (D*4 grad + 16 ids) + U_launch * 2*D*4   (row read + row write per DISTINCT row)
    # U_launch = distinct table rows the batch touches, counted by the kernel itself
    # (tbe_profile_read_rows).  Uniform ids over the Criteo table sizes hit the 11 tiny tables
    # thousands of times per row, so U is ~10 rows/sample, not F*L = 26: pricing the backward at
    # F*L rows would credit the kernel with bytes it never moves (it did read > 100 % of peak that way).
    rows_upd = ctypes.c_int64(0)
    lib.tbe_profile_read_rows(ctypes.byref(rows_upd))
    U_launch = rows_upd.value / max(args.steps, 1)  # distinct rows per step on this rank
    fwd_bytes = feat_units * args.global_batch * (D * 4 + 8 + 8 + D * 4)
    bwd_bytes = feat_units * args.global_batch * (D * 4 + 16) + U_launch * 2 * D * 4
    bwd_bytes_all_distinct = feat_units * args.global_batch * (D * 4 + 16 + 2 * D * 4)
    kern = {}
    for name, slot, nbytes in (("tbe_fwd_short_kernel", 0, fwd_bytes), ("bwd_update_kernel", 1, bwd_bytes),
                               ("tbe_backward whole call (fused: linearize+sort+update+fixup; two-phase: update+fixup)", 2, bwd_bytes),
                               ("tbe_backward_prepare (linearize+sort)", 3, 0.0)):
        tot_ms, n = prof[slot]
        if n:
            avg_ms = tot_ms / max(args.steps, 1)  # per step (one launch per step at N = 1)
            kern[name] = {"avg_us": avg_ms * 1e3, "launches": n, "algorithmic_MB": nbytes / 1e6,
                          "GB/s": nbytes / (avg_ms * 1e-3) / 1e9}
    # the dominant KERNEL (slots 0 and 1 are single kernels; slots 2 and 3 are multi-kernel spans)
    dom = max((k for k in kern if k in ("tbe_fwd_short_kernel", "bwd_update_kernel")),
              key=lambda k: kern[k]["avg_us"], default=None)
    roofline = None
    if dom:
        # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction) measured by a separate rocprofv3
        # --pmc run of the same kernels (profiles/r0N_pmc_traffic.json); only valid for the N = 1 shape.
        traffic = None
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{n}_pmc_traffic.json") for n in (3, 2, 1))
                         if os.path.exists(q)), "")
        if world == 1 and args.global_batch == 65536 and not args.zipf and not args.row_cap and pmc_path:
            pmc = json.load(open(pmc_path))
            hit = [v for k, v in pmc.items() if dom.split("(")[0] in k]
            if hit:
                traffic = hit[0]["hbm_total_MB"] * 1e6
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(kern[dom]["GB/s"], 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(kern[dom]["GB/s"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": (f"profiles/{os.path.basename(pmc_path)}: separate rocprofv3 --pmc passes of the same "
                                       "kernels at this shape (FETCH_SIZE x 2 + WRITE_SIZE), not collected in this run"
                                       if traffic is not None else None),
                    # the spec peak above is what `frac` is quoted against; this is what a 1-GiB device copy reaches on
                    # THIS box (bytes read + written over the best of 5), measured before the timed region
                    "peak_measured_copy_GBs": round(copy_GBs, 1) if copy_GBs else None,
                    "frac_of_measured_copy": round(kern[dom]["GB/s"] / copy_GBs, 4) if copy_GBs else None,
                    "avg_launch_us": round(kern[dom]["avg_us"], 1),
                    "distinct_rows_per_sample": round(U_launch / args.global_batch, 2),
                    "bwd_MB_if_all_rows_distinct": round(bwd_bytes_all_distinct / 1e6, 1),
                    "all": {k: {kk: round(vv, 1) for kk, vv in v.items()} for k, v in kern.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.ebc_torch import time_cpu_baseline

        r = time_cpu_baseline(CRITEO_1TB_ROWS, D, batches=tuple(dict.fromkeys((4096, args.global_batch))),
                              seconds_budget=args.cpu_seconds)
        big, small = r["per_batch"][args.global_batch], r["per_batch"][4096]
        cap_txt = f"tables capped at {r['row_cap']} rows (host RAM < 100 GB)" if r["row_cap"] else "full-size tables (84.85 GiB in host memory)"
        cpu = {"value": round(big["train_samples_per_s"], 1), "unit": "samples/s", "cores": r["cores"], "kind": "port",
               "fwd_only_value": round(big["fwd_samples_per_s"], 1),
               "value_min_max": [round(x, 1) for x in big["train_samples_per_s_min_max"]],
               "batch_4096": {"value": round(small["train_samples_per_s"], 1),
                              "fwd_only_value": round(small["fwd_samples_per_s"], 1)},
               "sample": (f"reference EmbeddingBagCollection design (26 x nn.EmbeddingBag sum, sparse=True + SGD; "
                          f"embedding part only, no MLPs), {cap_txt}, batch {args.global_batch}: median of "
                          f"{big['train_iters']} train + {big['fwd_iters']} fwd iterations (after one warm-up each), batch 4096: "
                          f"{small['train_iters']} + {small['fwd_iters']}; tables built in {r['build_s']:.1f} s (untimed)")}

    # ---- whole-step fractions and the binding resource (model-based, per rank) ------------------------
    # end to end: algorithmic embedding bytes per sample x samples/s over W x HBM peak, and the
    # dense side's 14.75 MFLOP per sample over W x the fp32 MFMA peak
    end_to_end = {
        "hbm_frac": round((BYTES_FWD + BYTES_BWD_SGD) * value / (world * HBM_PEAK_GBS * 1e9), 4),
        "hbm_bytes_per_sample": BYTES_FWD + BYTES_BWD_SGD,
        "mfma_f32_frac": round(MFLOP_PER_SAMPLE_TRAIN * 1e6 * value / (world * MFMA_F32_PEAK_TFLOPS * 1e12), 4),
        "mflop_per_sample": MFLOP_PER_SAMPLE_TRAIN,
    }
    n_tw_per_rank = [sum(1 for i in range(F) if kind[i] == r) for r in range(world)]
    n_rw_feats = sum(1 for i in range(F) if kind[i] == -1)
    # pooled all-to-all: a rank sends [B_local, D] per feature it holds (row-wise features: every rank holds a
    # shard) to EACH peer over that peer's own link, forward; the same volume comes back as gradients
    link_bytes = B_local * D * 4 * (max(n_tw_per_rank) + n_rw_feats) if world > 1 else 0
    xgmi_bytes_rank = 2 * (world - 1) * B_local * D * 4 * (n_tw_per_rank[rank] + n_rw_feats) if world > 1 else 0
    ids_bytes_rank = (world - 1) * B_local * 8 * (F - n_dp) // max(world, 1) + (world - 1) * B_local * 8 * n_rw_feats if world > 1 else 0
    est_ms = {
        "mfma_f32": B_local * MFLOP_PER_SAMPLE_TRAIN * 1e6 / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3,
        "hbm_embedding": (fwd_bytes + bwd_bytes) / (HBM_PEAK_GBS * 1e9) * 1e3,
        "xgmi_busiest_link": 2 * link_bytes / (XGMI_LINK_GBS * 1e9) * 1e3,
    }
    binding = {"resource": max(est_ms, key=est_ms.get), "est_ms_per_step_at_peak": {k: round(v, 3) for k, v in est_ms.items()},
               "xgmi_bytes_per_rank_per_step": int(xgmi_bytes_rank + 2 * ids_bytes_rank),
               "note": "time each resource would need at its peak rate for this rank's share of one step; the largest binds"}

    bad = sort_giveups != 0 or bounds_errors != 0 or not replicas_identical
    if rank == 0:
        if roofline is not None:
            roofline["end_to_end"] = end_to_end
        out = {
            "metric": "samples/sec Criteo-1TB DLRM batch 65536", "value": round(value, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            # per-step times from HIP events on the compute stream (rank 0), next to the wall-clock mean above
            "median_ms_per_step": round(median_ms, 3), "min_max_ms_per_step": [round(step_ms[0], 3), round(step_ms[-1], 3)],
            "higher_is_better": True, "scaling": "strong",
            # the reference's only published number is 5 497 159.68 samples/s on 8 x A100-40GB
            # (examples/dlrm/README.MD:45, real Criteo data, end to end) — comparable at N = 8 only
            "vs_baseline": round(value / 5497159.68, 3) if world == 8 and backend == "nccl" else None,
            "dtype": "f32", "data": "synthetic", "hip_graphs": hip_graphs, "hip_graphs_note": graphs_note,
            "explicit_backward_steps": int(getattr(train_model, "explicit_steps", 0)),
            "half_batch_steps": int(getattr(train_model, "half_batch_steps", 0)),
            "prefetched_lookups": int(getattr(train_model, "prefetched_lookups", 0)),
            "tuned_gemms": tuned,
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0,
            "backend": dist.get_backend() if dist.is_initialized() else None,
            "rehearsal": (f"{world} ranks on {torch.cuda.device_count()} GPU(s) over gloo, all-to-all staged through the host: "
                          "NOT a multi-GPU timing" if world > 1 and backend == "gloo" else None),
            "launcher": os.environ.get("TORCHREC_AMD_BENCH_LAUNCHER", "external torchrun" if world > 1 else "direct"),
            "env": {k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "TORCHREC_AMD_RW_INPUT_DIST",
                                               "TORCHREC_AMD_PREFETCH_LOOKUP", "TORCHREC_AMD_FUSED_BCE",
                                               "TORCHREC_AMD_REHEARSAL_LINK_US", "TORCHREC_AMD_RCCL_HIGH_PRIORITY",
                                               "TBE_STREAM_PROBE", "TORCHREC_AMD_HALF_BATCHES", "TORCHREC_AMD_WGRAD_LATE_LAYERS") if k in os.environ},
            "config": {"workload": "DLRM Criteo-1TB shape: 26 tables (177.9M rows, 84.85 GiB fp32, D=128), 13 dense, "
                                   "pooling factor 1, dense 512-256-128, over 1024-1024-512-256-1, fused exact SGD",
                       "global_batch": args.global_batch, "local_batch": B_local,
                       "parallelism": (f"mp{world}: {F - n_rw - n_dp} table-wise + {n_rw} row-wise + {n_dp} replicated "
                                       f"(data-parallel) tables, dense dp{world}"),
                       "plan": {"table_wise": F - n_rw - n_dp, "row_wise": n_rw, "data_parallel": n_dp,
                                "table_wise_per_rank": n_tw_per_rank, "row_wise_arg": args.row_wise,
                                "source": ("pinned mixed plan (BASELINE config 3, SURVEY.md §8d)" if args.row_wise < 0 and world > 1
                                           else "torchrec_amd planner" if args.row_wise <= 0 else "--row-wise")},
                       "ids": f"zipf({args.zipf})" if args.zipf else "uniform", "row_cap": args.row_cap or None,
                       "distinct_batches": args.num_batches, "seed": args.seed, "data_ranks": data_ranks,
                       "rw_input_dist": shard._rw_mode_active if n_rw else None},
            "checks": checks, "binding": binding, "roofline": roofline, "cpu_baseline": cpu,
        }
        # a run whose kernels gave up, saw out-of-range ids or let the replicas drift measured the wrong computation:
        # the line goes to stderr and the exit code says so
        print(json.dumps(out), file=sys.stderr if bad else sys.stdout, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if bad:
        print(f"[bench] INVALID RUN: sort give-ups {sort_giveups}, bounds-check errors {bounds_errors}, "
              f"dense replicas identical: {replicas_identical}", file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    _args = parse()
    if "WORLD_SIZE" not in os.environ and (_args.gpus > 1 or _args.spawn):
        sys.exit(launch_ranks(_args))  # before torch is imported: the parent never touches the GPU
    if os.environ.get("TORCHREC_AMD_BENCH_DRYRUN"):
        dry_run(_args)
    else:
        main(_args)
