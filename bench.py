#!/usr/bin/env python3
"""bench.py -- training images/sec of the ACL-GAN step (dis_update + gen_update) at 256x256.

  python bench.py --gpus N --steps K --warmup W [--dtype fp32|bf16|fp16] [--size S --batch B]

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks (torch.distributed.run; RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or -- when WORLD_SIZE is unset -- this script
re-executes itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`
and relays rank 0's JSON line.

Workload = BASELINE.json configs[1]: male2female architecture (configs/male2female.yaml), 256x256, fp32,
batch 8 PER GPU (weak scaling), synthetic U(-1,1) images resident in HBM before the timed region, reference
init statistics, z from a seeded CPU generator.  One "step" = one dis_update followed by one gen_update on the
same batch (zero_grad + forward + backward + [gradient all-reduce] + Adam each): value = global_batch / (t_dis + t_gen).
--dtype bf16 / fp16 (BASELINE configs[2] / [4]) runs the heavy convolutions on v_mfma_f32_32x32x16_{bf16,f16} with
fp32 accumulation, fp32 master weights and Adam, fp16 with dynamic loss scaling.

Timing: K steps bracketed by barrier + synchronize, max over ranks -> `value` / `ms_per_step` (the contract);
per-step HIP events on the launch stream give `config.ms_per_step_median` (SURVEY.md 8d defines the median).

config.other_configs (N = 1, headline workload only): after the timed region, 2 warm-up + 5 timed steps each of the other single-GPU shapes
BASELINE.json names -- 512x512 fp32 B=4 (configs[3]) and the per-GPU shapes of the two 8-GPU rows, 256x256 bf16 B=8 (configs[2]) and fp16 B=32
(configs[4]) -- so that the driver's own bench file carries them; they are not `value`.  config.lanes: HIP streams the independent branches of an
update run on (csrc/engine.hip "Lanes"; --lanes N, default 3).

Extra objects in the JSON line:
  roofline     MFMA-bound.  achieved = contract FLOPs (2.623 TFLOP per image at 256^2: NECESSARY conv+linear work of
               one dis+gen step) x images per step / event-timed step, against the matrix peak of the
               compute dtype (157.3 TFLOP/s fp32, 2500 bf16/fp16).  `executed_*`: the FLOPs the kernels really issue
               (sub-pixel path: the two upsample+5x5 layers with 9 instead of 25 taps away from the border; fp32: the 3x3
               ResBlock convolutions through Winograd F(4x4,3x3) with 1/4 of the MACs), i.e. the hardware utilisation
               figure -- the contract `frac` can approach or exceed 1 because of those algorithmic savings.
               "kernel": the dominant kernel alone, timed with HIP events.
  cpu_baseline the CPU oracle (a port of the reference step, oracle/aclgan_oracle.py) on the host cores, rank 0 at
               N=1 only: 1 warm-up + 3 timed steps at 256x256 B=1, median and spread.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

TFLOP_PER_IMAGE_256 = 2.623   # necessary work, dis+gen step, 256x256 (1311.6 GMAC)
UPCONV_GMAC_256 = 483.2       # of which the two "Upsample(2)+5x5" decoder layers: fwd x8 decodes, dgrad+wgrad x5
PEAK = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}   # TFLOP/s dense matrix peaks, MI355X_MICROARCH.md
CPU_THREADS_CAP = 32   # oneDNN/ATen on the GPU box's 256 hardware threads thrashes on B=1 tensors


def load_config(path=None):
    import yaml
    with open(path or os.path.join(ROOT, "configs", "male2female.yaml")) as f:
        return yaml.safe_load(f)


def male2female_config():
    return load_config()


# per-GPU batch of the BASELINE.json configuration each shipped YAML stands for (configs[1] / [2] / [3]; configs[4] = male2female fp16: 32)
BASELINE_BATCH = {"male2female": 8, "selfie2anime": 8, "glasses_removal": 4}


TRAFFIC_FILES = ("profiles/r06_step_traffic.json", "profiles/r05_step_traffic.json")      # newest first


def library_md5():
    import hashlib
    try:
        with open(os.path.join(ROOT, "acl-gan_amd", "libaclgan_hip.so"), "rb") as f:
            return hashlib.md5(f.read()).hexdigest()
    except OSError:
        return None


def step_traffic(dtype, S, B, launches_per_step=None):
    """memory-side bytes of ONE step from the committed PMC passes (rocprofv3 --pmc cannot run inside the timed process):
    profiles/r06_step_traffic.json (r05 as a fallback), written by scripts/step_traffic.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    runs of scripts/probe_step.py = sum over every kernel of one dis_update + gen_update of 2 x FETCH_SIZE (gfx950 correction,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE.  The entry records the build it was measured on (md5 of libaclgan_hip.so, kernel
    launches per step): when either differs from the library that is running, the figure is reported with stale = True.
    Returns (bytes or None, source, stale, entry)."""
    for tf in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, tf)) as f:
                ent = json.load(f).get("%s_%d_b%d" % (dtype, S, B))
            if not ent or ent.get("bytes_per_step") is None:
                continue
            stale = ent.get("lib_md5") != library_md5()
            if launches_per_step is not None and ent.get("launches_per_step") is not None:
                stale = stale or abs(float(ent["launches_per_step"]) - float(launches_per_step)) > 0.5
            return ent["bytes_per_step"], "%s:%s_%d_b%d" % (tf, dtype, S, B), bool(stale), ent
        except Exception:   # noqa: BLE001
            continue
    return None, None, None, None


def executed_flops(L, ctx, B, S):
    """matrix-pipe FLOPs ONE dis_update + gen_update executes at this shape, from the library's own dry run (aclgan_step_executed_flops: every
    convolution at the cost of the path its launchers choose -- direct, Winograd F(4x4,3x3) in whole tile blocks, sub-pixel phases + ring,
    parity phases of the stride-2 layers).  Round 6: replaces the closed formula of flops_per_image (kept below as a cross-check), which
    could not know which layers the cost models send to which kernel."""
    import ctypes as C
    tot = 0.0
    for which in (0, 1):
        v = C.c_double()
        L.check(L.lib.aclgan_step_executed_flops(ctx, which, B, S, S, C.byref(v)), "step_executed_flops")
        tot += v.value
    return tot


RESBLOCK_GMAC_256 = 637.8     # of which the 3x3 ResBlock convs: 120 forward + 72 dgrad + 72 wgrad launches x 2.4159 GMAC


def flops_per_image(S, dtype="fp32"):
    """(contract, executed) TFLOP per image of one dis+gen step at SxS.  Executed = what the kernels really issue: inside the
    border ring the sub-pixel decomposition runs 9/25 of the upsample+5x5 MACs, and (fp32 only) the Winograd F(4x4,3x3) path runs
    the 3x3 ResBlock convolutions -- forward, dgrad interior, wgrad -- and the four VALID 3x3 phases of the sub-pixel layers with
    36/144 of theirs (times the ragged-tile overhead of the (h-2) x (h-2) phase views)."""
    scale = (S / 256.0) ** 2
    interior = 0.5 * (((S // 2 - 4) / (S // 2)) ** 2 + ((S - 4) / S) ** 2)
    wino = dtype == "fp32" and S % 16 == 0 and os.environ.get("ACLGAN_NOWINO", "0") in ("", "0")
    phase_gmac = UPCONV_GMAC_256 * (9.0 / 25.0) * interior
    executed_gmac = 1311.6 - UPCONV_GMAC_256 * interior + phase_gmac
    if wino:
        executed_gmac -= RESBLOCK_GMAC_256 * 0.75
        if os.environ.get("ACLGAN_NOWINOUP5", "0") in ("", "0"):
            ragged = 0.5 * sum((4.0 * -(-(h - 2) // 4) / (h - 2)) ** 2 for h in (S // 4, S // 2))
            executed_gmac -= phase_gmac * (1.0 - 0.25 * ragged)
    return TFLOP_PER_IMAGE_256 * scale, 2e-3 * executed_gmac * scale


def cpu_baseline_worker():
    """runs in a subprocess (see cpu_baseline): times the CPU oracle and prints one JSON line."""
    from oracle import aclgan_oracle as O
    cores = min(CPU_THREADS_CAP, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = male2female_config()
    nets = O.test_nets(cfg, 0)
    g = torch.Generator().manual_seed(1)
    orc = O.OracleTrainer(cfg, nets=nets)

    def step(H, B):
        x_a = torch.rand(B, 3, H, H, generator=g) * 2 - 1
        x_b = torch.rand(B, 3, H, H, generator=g) * 2 - 1
        z = [torch.randn(B, 8, 1, 1, generator=g) for _ in range(6)]
        t0 = time.perf_counter()
        orc.dis_update(x_a, x_b, z[:3])
        orc.gen_update(x_a, x_b, z[3:])
        return time.perf_counter() - t0

    step(64, 1)                       # thread pool, oneDNN primitives
    step(256, 1)                      # warm-up at the measured shape (1 warm-up + 3 timed)
    ts = sorted(step(256, 1) for _ in range(3))
    print(json.dumps({"value": round(1.0 / ts[1], 4), "unit": "images/s", "cores": cores, "kind": "port",
                      "sample": "median of 3 steps (dis_update+gen_update) at 256x256, B=1, fp32, %d threads, after 1 warm-up step" % cores,
                      "seconds": [round(t, 2) for t in ts]}), flush=True)


def cpu_baseline(timeout_s=200):
    """The CPU oracle (a port of the reference step) on the host cores; bounded: a subprocess with a
    hard timeout so that the default bench run always finishes within minutes."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(min(CPU_THREADS_CAP, os.cpu_count() or 1))
    env["HIP_VISIBLE_DEVICES"] = ""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], env=env, capture_output=True,
                           text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:   # noqa: BLE001  (reported, never fatal for the GPU number)
        return {"value": None, "unit": "images/s", "cores": min(CPU_THREADS_CAP, os.cpu_count() or 1), "kind": "port",
                "sample": "CPU oracle steps at 256x256 B=1 did not finish within %ds (%s)" % (timeout_s, type(e).__name__)}


def dominant_kernel_probe(L, dtype, reps=20):
    """The step's dominant kernel on the ResBlock shape (B=8, 64x64, 256->256, 3x3), timed with HIP events on the launch stream.
    fp32: the 36 batched GEMM slices of the Winograd F(4x4,3x3) pipeline (conv_fwd_fast_kernel<2,2,1,2,4>, 20 % of the step) -- the launch
    ALONE through aclgan_gemm_slices_f32: frac = the FLOPs that launch issues / time / 157.3; the four-launch convolution it belongs to is
    reported beside it (`pipeline_*`: algorithmic = direct-convolution FLOPs).  bf16 / fp16: conv_fwd16 (one launch = the convolution)."""
    import ctypes as C
    B, H, Cc = 8, 64, 256
    flop_direct = 2.0 * (B * H * H) * Cc * (9 * Cc)
    x = torch.randn(B, H, H, Cc, device="cuda")
    w = torch.randn(Cc, 3, 3, Cc, device="cuda") * 0.02
    b = torch.zeros(Cc, device="cuda")
    y = torch.empty(B, H, H, Cc, device="cuda")
    d = L.ConvDesc(B, H, H, Cc, Cc, 3, 1, 1, 0, 0)
    st = L.stream_ptr()

    def timed(call):
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    if dtype == "fp32":
        # the step's dominant kernel (round 4): the ResBlock convolution as ONE launch -- csrc/conv_wino_fused.hip, 196 launches and ~20 % of the
        # fp32 step.  Timed ALONE (the filter transform is cached per update in the step): frac = the FLOPs the launch issues on the matrix
        # pipe (36 frequency GEMMs = 1/4 of the direct convolution) / time / 157.3; `algorithmic_*` prices the direct convolution it replaces.
        T = B * (H // 4) * (H // 4)
        Uf = torch.empty(36 * Cc * Cc, device="cuda")
        L.check(L.lib.aclgan_winograd_filter_frag(L.ptr(w), L.ptr(Uf), Cc, Cc, 0, st))
        ms = timed(lambda: L.check(L.lib.aclgan_conv3x3_winograd_fused(L.ptr(x), L.ptr(Uf), L.ptr(b), L.ptr(y), B, H, H, Cc, Cc, 0, 1, 0, None, st)))
        flop = 2.0 * 36 * T * Cc * Cc
        alg_bytes = 4.0 * (x.numel() + y.numel() + 36 * Cc * Cc)      # x read, y written, U read: what the launch must move
        pmc, pmc_file = None, None
        for pf in ("r06_pmc_wino_fused.json", "r05_pmc_wino_fused.json"):      # newest first
            try:
                with open(os.path.join(ROOT, "profiles", pf)) as f:
                    pmc, pmc_file = json.load(f), pf
                break
            except Exception:   # noqa: BLE001
                continue
        out = {"name": "wino_fused_kernel: ResBlock conv 8x64x64x256->256 3x3 reflect-pad, Winograd F(4x4,3x3) input transform + 36 GEMMs [2048 x 256] x [256 x 256] + "
                       "output transform in one launch",
               "ms": round(ms, 4), "flop_per_launch": flop, "achieved": round(flop / ms / 1e9, 2), "unit": "TFLOP/s",
               "frac": round(flop / ms / 1e9 / PEAK[dtype], 4),
               "algorithmic_flop_per_launch": flop_direct, "algorithmic_achieved": round(flop_direct / ms / 1e9, 2),
               "algorithmic_frac": round(flop_direct / ms / 1e9 / PEAK[dtype], 4),
               "algorithmic_bytes": alg_bytes,
               "traffic": None if not pmc else pmc.get("bytes_per_launch"),
               "traffic_source": None if not pmc else "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE of this launch)" % pmc_file,
               "traffic_stale": None if not pmc else bool(pmc.get("lib_md5") != library_md5()),
               "replaces": "round 3: wino_input + 36-slice GEMM launch + wino_output = 157 us and 469 MB of memory-side traffic per convolution"}
        return out
    code = L.DTYPE[dtype]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    w16 = torch.empty(w.numel(), dtype=torch.int16, device="cuda")
    L.check(L.lib.aclgan_pack_weights16(L.ptr(w), L.ptr(w16), None, Cc, 9, Cc, code, st))
    if L.lib.aclgan_conv16s_ok(C.byref(d), 0) and os.environ.get("ACLGAN_ACT16", "1") not in ("0",):
        # the ResBlock convolution as the 16-bit step runs it: activations stored in the compute dtype, operand tiles global -> LDS
        x16 = x.to(tdt); y16 = torch.empty(B, H, H, Cc, device="cuda", dtype=tdt)
        ms = timed(lambda: L.check(L.lib.aclgan_conv2d_fwd16s(C.byref(d), code, L.ptr(x16), L.ptr(w16), L.ptr(b), L.ptr(y16), code, st)))
        name = "conv_fwd16s_kernel<%s,128> 8x64x64x256->256 3x3 (ResBlock conv; %s activations in HBM, LDS-DMA operand tiles)" % (dtype, dtype)
        alg = 2.0 * (x.numel() + y.numel() + w.numel())
    else:
        ms = timed(lambda: L.check(L.lib.aclgan_conv2d_fwd16(C.byref(d), code, L.ptr(x), L.ptr(w), L.ptr(w16), L.ptr(b), L.ptr(y), None, st)))
        name = "conv_fwd16_kernel<%s> 8x64x64x256->256 3x3 (ResBlock conv, fp32 activations in HBM)" % dtype
        alg = 4.0 * (x.numel() + y.numel()) + 2.0 * w.numel()
    return {"name": name, "ms": round(ms, 4), "flop_per_launch": flop_direct, "achieved": round(flop_direct / ms / 1e9, 2), "unit": "TFLOP/s",
            "frac": round(flop_direct / ms / 1e9 / PEAK[dtype], 4), "traffic": None, "algorithmic_bytes": alg}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run and relay
    rank 0's output (the one JSON line)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=str, default=None, help="YAML under configs/ naming the workload (default configs/male2female.yaml = BASELINE configs[1]; "
                    "selfie2anime.yaml = configs[2], glasses_removal.yaml = configs[3]): architecture, hyper-parameters, image size and compute_dtype come from it")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch; default: the BASELINE batch of the config (8 / 8 / 4), 32 with --dtype fp16 (configs[4]: 256 over 8 GPUs)")
    ap.add_argument("--size", type=int, default=None, help="image size; default: the config's crop_image_height")
    ap.add_argument("--dtype", choices=["fp32", "bf16", "fp16"], default=None, help="compute dtype of the heavy convolutions; default: the config's compute_dtype (fp32)")
    ap.add_argument("--ddp-overlap", type=int, choices=[0, 1], default=None, help="N > 1: 1 (default) = gradient buckets all-reduced from inside the backward; "
                    "0 = plain bucketed all-reduce after the backward (fallback if the overlapped path misbehaves on a new RCCL / topology)")
    ap.add_argument("--deterministic", action="store_true", help="ordered reductions everywhere (bit-reproducible step); default: the fast plan")
    ap.add_argument("--graph", action="store_true", help="replay each update from a captured HIP graph (launch-bound regimes: small batches / images)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-floor", action="store_true", help="skip the 64x64 B=1 launch-bound probe (keeps kernel traces clean)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip config.other_configs (the other single-GPU shapes of BASELINE.json, 5 steps each, after the timed region)")
    ap.add_argument("--lanes", type=int, default=None, help="streams the independent branches of an update are spread over (1 .. 3; default: the library's, ACLGAN_LANES or 3; 1 = one queue)")
    ap.add_argument("--pre-streams", type=int, default=0, help="experiment (round 6): create this many HIP streams and run one kernel on each BEFORE the library creates "
                    "its lane streams -- what a data-parallel process group (RCCL's stream) or a prefetching loader does to HIP's stream -> hardware-queue placement")
    ap.add_argument("--post-streams", type=int, default=0, help="experiment (round 6): ... or AFTER the trainer exists and before its first update (what a process group "
                    "initialised by the trainer's rank-0 broadcast does)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker()
        return

    cfg_path = args.config or os.path.join(ROOT, "configs", "male2female.yaml")
    cfg = load_config(cfg_path)
    cfg_name = os.path.splitext(os.path.basename(cfg_path))[0]
    if args.dtype is None:
        args.dtype = str(cfg.get("compute_dtype", "fp32"))
    if args.size is None:
        args.size = int(cfg.get("crop_image_height", 256))
    if args.batch is None:
        args.batch = 32 if (args.dtype == "fp16" and cfg_name == "male2female") else BASELINE_BATCH.get(cfg_name, int(cfg.get("batch_size", 8)))
    if args.ddp_overlap is not None:
        os.environ["ACLGAN_DDP_OVERLAP"] = str(args.ddp_overlap)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")    # a failed / hung collective tears the job down instead of hanging the bench
    os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "1")
    t_start = time.perf_counter()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback for the product path")
    # test hooks: ACLGAN_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and ACLGAN_DIST_BACKEND=gloo replaces RCCL (which refuses two
    # ranks on one device), so that the complete N>1 control flow can be exercised on a 1-GPU box (tests/test_gpu_ddp.py)
    if os.environ.get("ACLGAN_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit("rank %d: local GPU %d not visible (%d devices)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # Round 6: the lane scheduler's streams BEFORE the process group exists.  HIP binds streams to its hardware queues in creation order;
    # init_process_group(device_id=...) initialises RCCL's communicator (and its streams) eagerly, and a foreign stream created before the
    # lanes' costs the 3-lane step 2.5 - 3.7 ms (profiles/r06_experiments.md section 8: --pre-streams / --post-streams).
    import aclgan_amd  # noqa: F401  (raises if libaclgan_hip.so is missing)
    from aclgan_amd import _lib as L
    if args.lanes is not None:
        L.check(L.lib.aclgan_tuning(b"lanes", args.lanes, None), "tuning lanes")
    if os.environ.get("ACLGAN_WARM_STREAMS", "1") not in ("", "0") and not args.pre_streams:
        L.check(L.lib.aclgan_warm_streams(0), "warm_streams")
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("ACLGAN_BENCH_FORCE_DIST") == "1"   # the latter: exercise the RCCL path on one GPU
    rccl_log = None
    if use_dist and os.environ.get("ACLGAN_DIST_BACKEND", "nccl") == "nccl" and os.environ.get("ACLGAN_BENCH_RCCL_LOG", "1") != "0":
        # first contact with a multi-GPU node: keep what RCCL decided (topology, algorithm / protocol per message size) in a per-rank file and
        # quote it on the JSON line -- never on stdout
        rccl_log = "/tmp/aclgan_rccl_%d_rank%d.log" % (os.getppid(), rank)
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,COLL")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("ACLGAN_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from aclgan_amd.trainer import aclgan_Trainer
    import ctypes as C

    def tuning_value(key):      # (read a switch; the getter changes nothing)
        v = C.c_longlong()
        L.check(L.lib.aclgan_tuning_get(key, C.byref(v)), "tuning_get")
        return int(v.value)

    cfg["display_size"] = 1
    if os.environ.get("ACLGAN_BENCH_TEST_WIDTH"):      # test hook (tests/test_gpu_ddp.py): the control flow of N ranks sharing one GPU over gloo, with
        tw = int(os.environ["ACLGAN_BENCH_TEST_WIDTH"])     # networks narrow enough that gloo's host-side all-reduce does not dominate the test's wall time
        cfg["gen"].update(dim=tw, mlp_dim=2 * tw); cfg["dis"].update(dim=tw)
    torch.manual_seed(0)       # (replicas are made identical by the trainer's rank-0 broadcast, not by this seed)
    pre_streams = []
    for _ in range(max(0, args.pre_streams)):      # (kept alive for the whole run; each has seen work, so HIP has bound it to a hardware queue)
        ps = torch.cuda.Stream(device="cuda:%d" % local_rank)
        with torch.cuda.stream(ps):
            torch.zeros(1024, device="cuda:%d" % local_rank).add_(1.0)
        pre_streams.append(ps)
    torch.cuda.synchronize()
    tr = aclgan_Trainer(cfg, device="cuda:%d" % local_rank, compute_dtype=args.dtype, deterministic=True if args.deterministic else None,
                        hip_graph=True if args.graph else None)
    for _ in range(max(0, args.post_streams)):
        ps = torch.cuda.Stream(device="cuda:%d" % local_rank)
        with torch.cuda.stream(ps):
            torch.zeros(1024, device="cuda:%d" % local_rank).add_(1.0)
        pre_streams.append(ps)
    torch.cuda.synchronize()
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1 + rank)   # each rank its own shard of the synthetic global batch
    x_a = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    x_b = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    zs = [[torch.randn(B, 8, 1, 1, generator=g) for _ in range(3)] for _ in range(2)]

    def step():
        tr.dis_update(x_a, x_b, cfg, z=zs[0])
        tr.gen_update(x_a, x_b, cfg, z=zs[1])
        tr.update_learning_rate()

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    log("trainer built (%s, world %d); warm-up" % (args.dtype, world))
    overlap_fallback = None
    for w_i in range(args.warmup):
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            # first contact hardening: if the overlapped bucket reducer fails on this RCCL / topology, fall back ONCE to the plain bucketed
            # all-reduce after the backward (every rank takes the same branch: the failure modes seen so far -- an unsupported async
            # option, a callback raising -- are deterministic across ranks) and rebuild the trainer
            if not (use_dist and getattr(tr, "_reducer", None) is not None and overlap_fallback is None):
                raise
            overlap_fallback = repr(e)[:300]
            log("overlapped all-reduce failed (%s): falling back to --ddp-overlap 0" % overlap_fallback)
            os.environ["ACLGAN_DDP_OVERLAP"] = "0"
            tr = aclgan_Trainer(cfg, device="cuda:%d" % local_rank, compute_dtype=args.dtype, deterministic=True if args.deterministic else None,
                                hip_graph=True if args.graph else None)
            step()
            torch.cuda.synchronize()
        log("warm-up step done")
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    launches0 = L.lib.aclgan_launch_count()
    tr.allreduce_exposed_ms()      # reset
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    t_enq = time.perf_counter() - t0      # every launch of the K steps is queued here; the GPU is still working
    launches_per_step = (L.lib.aclgan_launch_count() - launches0) / float(args.steps)
    exposed_ms = 0.0
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    exposed_ms = tr.allreduce_exposed_ms() / args.steps      # compute-stream time spent waiting for gradient collectives, per step
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    ev_ms = sum(per_step)
    if use_dist:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        te = torch.tensor([exposed_ms], device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        exposed_ms = float(te.item())
    losses_ok = all(map(lambda n: torch.isfinite(getattr(tr, n)).item(), ["loss_gen_total", "loss_dis_total"]))
    rccl_info = None
    if use_dist:
        rccl_info = {"overlap_fallback": overlap_fallback, "log": rccl_log,
                     # which queue a bucket's all-reduce is ordered after (csrc/engine.hip run_tape): ProcessGroupNCCL records its event on the
                     # caller's current stream = lane 0, and the engine makes lane 0 wait for every lane and for the parameter-gradient
                     # stream before it fires the bucket callback
                     "collectives_ordered_after": "lane 0 (caller's stream) after it joined all lanes and the parameter-gradient stream"}
        try:
            if rccl_log and os.path.exists(rccl_log):
                lines = open(rccl_log, errors="replace").read().splitlines()
                pick = [l.split("NCCL INFO", 1)[-1].strip() for l in lines if any(k in l for k in ("Algo", "Proto", "Ring ", "Tree ", "Channel", "Connected all", "comm "))]
                seen, uniq = set(), []
                for l in pick:
                    k = l[:60]
                    if k not in seen:
                        seen.add(k); uniq.append(l[:160])
                rccl_info["decisions"] = uniq[:24]
        except Exception as e:      # noqa: BLE001
            rccl_info["decisions_error"] = repr(e)[:200]
    replicas_identical = None
    if use_dist:      # (outside the timed region) data-parallel replicas must still hold identical parameters after K steps
        chk = torch.stack([tr._param[0].double().sum(), tr._param[1].double().sum(), tr._param[0].double().abs().sum()])
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        replicas_identical = all(bool(torch.equal(lst[0], x)) for x in lst)

    # secondary figure: the reference loop's cadence D_update=1, G_update=2 -> B / (t_dis + t_gen / 2);
    # two extra steps OUTSIDE the timed region, split with events between the two updates
    ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_dis = t_gen = 0.0
    for _ in range(2):
        ev3[0].record(); tr.dis_update(x_a, x_b, cfg, z=zs[0])
        ev3[1].record(); tr.gen_update(x_a, x_b, cfg, z=zs[1])
        ev3[2].record(); torch.cuda.synchronize()
        t_dis += ev3[0].elapsed_time(ev3[1]) / 2; t_gen += ev3[1].elapsed_time(ev3[2]) / 2

    # launch-bound floor (outside the timed region, rank 0): the same step on 64x64 B=1 images -- the same ~2000 launches and the
    # same host-side tape, next to no GPU work -- costs what the host needs to enqueue a step; this is what a HIP graph would remove
    launch_floor_ms = None
    if world == 1 and not args.no_launch_floor:      # (a second trainer would enter the data-parallel broadcast on one rank only)
        try:
            g2 = torch.Generator().manual_seed(3)
            xs = torch.rand(1, 3, 64, 64, generator=g2).cuda() * 2 - 1
            z1 = [torch.randn(1, cfg["gen"]["style_dim"], 1, 1, generator=g2) for _ in range(3)]
            tr2 = aclgan_Trainer(cfg, device="cuda:%d" % local_rank, compute_dtype=args.dtype, hip_graph=True if args.graph else None)
            for i in range(6):
                if i == 2:
                    torch.cuda.synchronize(); tf0 = time.perf_counter()
                tr2.dis_update(xs, xs, cfg, z=z1); tr2.gen_update(xs, xs, cfg, z=z1)
            torch.cuda.synchronize()
            launch_floor_ms = (time.perf_counter() - tf0) * 1e3 / 4
            del tr2
        except Exception as e:      # informational only
            log("launch floor probe failed: %r" % (e,))

    # the reference's own batch size (configs/male2female.yaml:13 `batch_size: 3`; BASELINE quotes the metric at 8): the launch-bound end of
    # the regime, outside the timed region, rank 0 at N=1 only
    small_batch = None
    if world == 1 and not args.no_launch_floor and S == 256:
        try:
            Bs = int(load_config(cfg_path).get("batch_size", 3))
            if Bs != B:
                g3 = torch.Generator().manual_seed(5)
                xsa = (torch.rand(Bs, 3, S, S, generator=g3) * 2 - 1).cuda(); xsb = (torch.rand(Bs, 3, S, S, generator=g3) * 2 - 1).cuda()
                z3 = [torch.randn(Bs, cfg["gen"]["style_dim"], 1, 1, generator=g3) for _ in range(3)]
                tr3 = aclgan_Trainer(cfg, device="cuda:%d" % local_rank, compute_dtype=args.dtype, hip_graph=True if args.graph else None)
                n3 = 6
                for i in range(2 + n3):
                    if i == 2:
                        torch.cuda.synchronize(); ts0 = time.perf_counter(); l0 = L.lib.aclgan_launch_count()
                    tr3.dis_update(xsa, xsb, cfg, z=z3); tr3.gen_update(xsa, xsb, cfg, z=z3)
                torch.cuda.synchronize()
                ms3 = (time.perf_counter() - ts0) * 1e3 / n3
                small_batch = {"batch": Bs, "why": "the reference's own batch_size (configs/male2female.yaml:13); not the BASELINE metric's configuration",
                               "ms_per_step": round(ms3, 3), "images_per_s": round(Bs / ms3 * 1e3, 2),
                               "kernel_launches_per_step": round((L.lib.aclgan_launch_count() - l0) / float(n3), 1)}
                del tr3
        except Exception as e:      # informational only
            log("small-batch probe failed: %r" % (e,))

    # the other single-GPU shapes BASELINE.json names (configs[3] 512x512 fp32 B=4; the per-GPU shapes of the two 8-GPU rows: configs[2]
    # 256x256 bf16 B=8, configs[4] 256x256 fp16 B=32), 2 warm-up + 5 timed steps each, AFTER the timed region of the headline workload and only
    # when the run IS the headline workload: so that the driver's own bench file carries them (they are not `value`)
    other_configs = None
    headline = (world == 1 and cfg_name == "male2female" and args.dtype == "fp32" and S == 256 and B == 8 and not args.deterministic and not args.graph
                and not os.environ.get("ACLGAN_BENCH_TEST_WIDTH"))
    if headline and not args.no_other_configs and os.environ.get("ACLGAN_BENCH_OTHER", "1") != "0":
        other_configs = []
        tr._ws = None; tr._ws_shape = None      # (the headline trainer is not stepped again: give its arena back before the larger shapes allocate theirs)
        torch.cuda.empty_cache()
        for (yaml_name, odt, oS, oB, label) in (("glasses_removal", "fp32", 512, 4, "configs[3]: glasses-removal 512x512 fp32, batch=4, 1 GPU"),
                                                ("selfie2anime", "bf16", 256, 8, "configs[2] per-GPU shape: selfie2anime 256x256 bf16, batch=8 (global 64 over 8 GPUs)"),
                                                ("male2female", "fp16", 256, 32, "configs[4] per-GPU shape: male2female 256x256 fp16 MFMA + loss scaling, batch=32 (global 256 over 8 GPUs)")):
            try:
                ocfg = load_config(os.path.join(ROOT, "configs", yaml_name + ".yaml"))
                ocfg["display_size"] = 1
                otr = aclgan_Trainer(ocfg, device="cuda:%d" % local_rank, compute_dtype=odt)
                go = torch.Generator().manual_seed(11)
                oa = (torch.rand(oB, 3, oS, oS, generator=go) * 2 - 1).cuda(); ob = (torch.rand(oB, 3, oS, oS, generator=go) * 2 - 1).cuda()
                oz = [torch.randn(oB, ocfg["gen"]["style_dim"], 1, 1, generator=go) for _ in range(3)]
                n_o = 5
                for i in range(2 + n_o):
                    if i == 2:
                        torch.cuda.synchronize(); to0 = time.perf_counter(); lo0 = L.lib.aclgan_launch_count()
                    otr.dis_update(oa, ob, ocfg, z=oz); otr.gen_update(oa, ob, ocfg, z=oz); otr.update_learning_rate()
                torch.cuda.synchronize()
                oms = (time.perf_counter() - to0) * 1e3 / n_o
                oexec = executed_flops(L, otr._ctx, oB, oS) / 1e12 / oB      # TFLOP per image, the library's own count
                other_configs.append({"workload": label, "config_file": "configs/%s.yaml" % yaml_name, "dtype": odt, "size": oS, "batch": oB, "steps": n_o, "warmup": 2,
                                      "ms_per_step": round(oms, 3), "images_per_s": round(oB / oms * 1e3, 2),
                                      "frac": round(oexec * oB / (oms / 1e3) / PEAK[odt], 4), "peak_TFLOPs": PEAK[odt],
                                      "kernel_launches_per_step": round((L.lib.aclgan_launch_count() - lo0) / float(n_o), 1),
                                      "losses_finite": bool(torch.isfinite(otr.loss_gen_total).item() and torch.isfinite(otr.loss_dis_total).item())})
                del otr, oa, ob
                torch.cuda.empty_cache()
            except Exception as e:      # informational only: never fatal for the headline number
                other_configs.append({"workload": label, "error": repr(e)[:300]})
                log("other-config probe %s failed: %r" % (label, e))
        log("other configs done")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * B / (elapsed / args.steps)
        tflop_img, tflop_formula = flops_per_image(S, args.dtype)
        flop_model = executed_flops(L, tr._ctx, B, S)      # FLOPs per step (per GPU) as the kernels execute them: the library's dry run
        tflop_exec = flop_model / 1e12 / B
        step_s = ev_ms / args.steps / 1e3
        ach = tflop_img * B / step_s      # per GPU, from HIP events on the launch stream
        peak = PEAK[args.dtype]
        name = cfg_name if (args.config or args.dtype == "fp32") else {"bf16": "selfie2anime (male2female architecture)", "fp16": "male2female"}[args.dtype]
        # memory side of the step: algorithmic bytes from a dry run of the scheduler (every operator's inputs read once, outputs written
        # once), measured bytes from the committed PMC passes
        alg_bytes = 0.0
        for which in (0, 1):
            v = C.c_double()
            L.check(L.lib.aclgan_step_algorithmic_bytes(tr._ctx, which, B, S, S, C.byref(v)), "step_algorithmic_bytes")
            alg_bytes += v.value
        traffic, traffic_src, traffic_stale, traffic_ent = step_traffic(args.dtype, S, B, launches_per_step)
        ex_ach = tflop_exec * B / step_s
        # ... and the hardware's own count of the same step where a PMC pass of THIS build is committed (scripts/step_mfma_flops.py)
        flop_counter = None if not traffic_ent else traffic_ent.get("mfma_flop_per_step")
        flop_counter_stale = None if flop_counter is None else bool(traffic_ent.get("mfma_lib_md5") != library_md5())
        out = {
            "metric": "training images/sec at 256x256 (gen+dis step)" if S == 256 else "training images/sec at %dx%d (gen+dis step)" % (S, S),
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic U(-1,1) A/B batches, reference init statistics, seeded z",
            "config": {"workload": "%s %dx%d %s, batch=%d per GPU: dis_update + gen_update (fwd+bwd+Adam each)%s" %
                                   (name, S, S, args.dtype, B, " [TEST WIDTH %s: not a benchmark]" % os.environ["ACLGAN_BENCH_TEST_WIDTH"] if os.environ.get("ACLGAN_BENCH_TEST_WIDTH") else ""),
                       "config_file": os.path.relpath(cfg_path, ROOT),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "rccl_world_size": dist.get_world_size() if use_dist else 1,
                       "dist_backend": dist.get_backend() if use_dist else None, "replicas_identical": replicas_identical,
                       "grad_allreduce": ("overlapped with backward (bucket callback)" if getattr(tr, "_reducer", None) is not None else
                                          ("after backward" if use_dist else "none (1 GPU)")),
                       # compute-stream time per step spent WAITING for the gradient exchange (events around finish() of the bucket reducer /
                       # the post-backward all-reduce), max over ranks: what of the all-reduce was NOT hidden behind the backward
                       "ms_allreduce_exposed": round(exposed_ms, 3) if use_dist else None,
                       "deterministic": bool(tr.deterministic), "hip_graph": bool(tr.hip_graph and tr._graphs.get("gen", {}).get("graph") is not None),
                       "losses_finite": bool(losses_ok), "ms_per_step_median": round(statistics.median(per_step), 3),
                       "ms_per_step_min_max": [round(min(per_step), 3), round(max(per_step), 3)],
                       "ms_dis_update": round(t_dis, 2), "ms_gen_update": round(t_gen, 2),
                       # host time until the K steps were queued (the HIP queue back-pressures, so this tracks the GPU when it is the
                       # bottleneck) and the launch-bound floor: the whole step on 64x64 B=1 images, i.e. the same launches with next to no work
                       "host_enqueue_ms_per_step": round(t_enq * 1e3 / args.steps, 2),
                       "kernel_launches_per_step": round(launches_per_step, 1),
                       "launch_bound_floor_ms_per_step": None if launch_floor_ms is None else round(launch_floor_ms, 2),
                       "reference_cadence_D1_G2_images_per_s": round(world * B / ((t_dis + 0.5 * t_gen) / 1e3), 2),
                       "small_batch": small_batch, "other_configs": other_configs,
                       # scheduler switches of this run (csrc/engine.hip): branches of an update on `lanes` HIP streams; in a data-parallel run the
                       # gradient buckets are handed to RCCL from lane 0 = the caller's stream AFTER it has joined every lane and the
                       # parameter-gradient stream
                       "lanes": tuning_value(b"lanes"), "batched_filter_transforms": bool(tuning_value(b"u_batch")),
                       "pre_streams": len(pre_streams),
                       "rccl": rccl_info},
            # frac = what the MFMA pipes really issue (EXECUTED FLOPs: Winograd F(4x4,3x3) runs the 3x3 convolutions with 1/4 of the
            # direct-convolution MACs, the sub-pixel path the upsample+5x5 layers with 9/25) over the dense matrix peak: a hardware
            # fraction, never above 1.  algorithmic_* = the SURVEY 8d contract figure (direct-convolution FLOPs of the step): it can
            # exceed 1 exactly because of those two algebraic reductions.
            "roofline": {"bound": "mfma", "achieved": round(ex_ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ex_ach / peak, 4),
                         "flop_per_launch": flop_model, "launch": "one dis_update+gen_update step (per GPU)",
                         "flop_source": "aclgan_step_executed_flops (dry run of the scheduler: the path every layer's launchers choose)",
                         "flop_per_launch_counter": flop_counter, "flop_counter_source": None if flop_counter is None else "rocprofv3 --pmc SQ_INSTS_MFMA x FLOPs per instruction, " + str(traffic_src),
                         "flop_counter_stale": flop_counter_stale,
                         "flop_counter_vs_model": None if not flop_counter else round(flop_counter / flop_model, 4),
                         "flop_mismatch": None if not flop_counter else bool(abs(flop_counter / flop_model - 1.0) > 0.02),
                         "flop_per_launch_formula_r05": tflop_formula * B * 1e12,
                         "event_ms_per_step": round(ev_ms / args.steps, 3),
                         "algorithmic_flop_per_launch": tflop_img * B * 1e12, "algorithmic_achieved": round(ach, 2),
                         "algorithmic_frac": round(ach / peak, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "traffic_build": None if not traffic_ent else {k: traffic_ent.get(k) for k in ("head", "lib_md5", "launches_per_step")},
                         "algorithmic_bytes": alg_bytes,
                         "traffic_ratio": None if not traffic else round(traffic / alg_bytes, 2),
                         "hbm_floor_ms": round(alg_bytes / 8e12 * 1e3, 2),
                         "note": ("frac = EXECUTED FLOPs / time / dense MFMA peak (hardware utilisation); algorithmic_frac = SURVEY 8d contract FLOPs "
                                  "(direct convolution, 2.623 TFLOP per image at 256x256) / time / peak -- above frac because Winograd F(4x4,3x3) and the "
                                  "sub-pixel decomposition execute fewer MACs than the contract counts; traffic = 2 x FETCH_SIZE + WRITE_SIZE summed over "
                                  "one step (committed PMC passes), algorithmic_bytes = every operator's inputs + outputs once")},
        }
        if args.dtype != "fp32":
            out["config"]["precision"] = ("heavy convolutions: %s operands on v_mfma_f32_32x32x16, fp32 accumulate; fp32 master weights, "
                                          "Adam, norm statistics and losses%s" % (args.dtype, "; dynamic loss scaling" if args.dtype == "fp16" else ""))
            if args.dtype == "fp16":
                out["config"]["loss_scale"] = tr.loss_scale_state()
        log("timed region done: %.1f ms/step" % ms_per_step)
        out["roofline"]["kernel"] = dominant_kernel_probe(L, args.dtype)
        log("kernel probe done; cpu baseline next")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio (block-buffered when stdout is a file): flush it on every rank first,
    # so that rank 0's JSON line is the LAST line of the job's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if rank == 0:
        time.sleep(0.2 if world > 1 else 0.0)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
