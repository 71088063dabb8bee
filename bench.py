#!/usr/bin/env python3
"""
Headline benchmark: denoised 64^3 PET sub-volumes per second at 250 DDPM
steps (BASELINE.json), published architecture, fp32, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch: p_sample_loop of
`--ddpm-steps` (250) reverse steps on `--batch` (1) volume(s) of 1x64^3 per
GPU, i.e. 250 UNet forwards + 250 fused sampler updates, device RNG included,
inputs resident in HBM.  Ranks work on independent volumes (weak scaling, no
collective inside a sample; one all_gather of the finished sample per step, as
scripts/test.py:74-78 does).  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself
(a torch.distributed.run child process; this parent never touches a GPU), relays
rank 0's line and exits with the child's status -- the reference's launcher is one
command too (test_DDPM_3d_mpi.sh:5).

roofline     : the conv3d implicit-GEMM kernel family (100 % of the path's
               FLOPs).  HIP events around every conv launch of one UNet
               forward per timed step; achieved = algorithmic FLOPs / event time.
               device_sustained_tflops = what THIS board's matrix pipes sustain in a
               register-only MFMA loop of the instruction the dominant kernel issues
               (ddpm3d_mfma_probe, ~50 ms, outside the timed region): boards of one
               pool differ by several per cent, frac_of_sustained does not.
parity       : the metric's "PSNR vs ref".  (a) the first timed volume against the same
               volume (same noise) in the exact-fp32 arithmetic of this library, all 250
               steps; (b) the first `--cpu-steps`+1 reverse steps against the CPU oracle
               (oracle/, pinned to the reference's own outputs) on the same injected noise.
cpu_baseline : oracle/ (the CPU restatement pinned to the reference) timed on
               this box's host cores on a bounded sample (a few p_sample steps
               of the same workload), extrapolated to volumes/s.  N=1, rank 0.
"""

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TRAFFIC_EPOCH = "wz8x4x4"   # see roofline.traffic_signature
PUBLISHED = dict(large_size=96, small_size=96, num_channels=128, num_res_blocks=2, num_head_channels=64,
                 attention_resolutions="1000", learn_sigma=True, resblock_updown=True,
                 use_scale_shift_norm=True)
TINY = dict(PUBLISHED, num_channels=32, num_res_blocks=1)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: f16/bf16 MFMA, dense
# Roofline peak per kernel family = MFMA peak of the instruction it issues / MFMAs it issues per
# ALGORITHMIC product (achieved is always priced in the direct form's FLOPs):
#   p0 exact fp32 MFMA 1;  p1 f16x3 direct 3;  p2 f16 direct 1;
#   p3 f16x3 on the Winograd-D form 3 * 2/3 = 2;  p4 f16 on the Winograd-D form 2/3.
PEAK_OF_TAG = {
    "p0": (PEAK_F32_MFMA_TFLOPS, "fp32 MFMA dense (157.3), 1 MFMA per product"),
    "p1": (PEAK_F16_MFMA_TFLOPS / 3.0, "f16 MFMA dense (2500) / 3 MFMAs per algorithmic fp32 product"),
    "p2": (PEAK_F16_MFMA_TFLOPS, "f16 MFMA dense (2500), 1 MFMA per product"),
    "p3": (PEAK_F16_MFMA_TFLOPS / 2.0, "f16 MFMA dense (2500) / 2 MFMAs per algorithmic fp32 product "
                                       "(3 split products x 2/3 Winograd F(2,3) along depth)"),
    "p4": (PEAK_F16_MFMA_TFLOPS * 1.5, "f16 MFMA dense (2500) x 3/2 (Winograd F(2,3) along depth issues "
                                       "2/3 MFMA per algorithmic product)"),
    "p5": (PEAK_F16_MFMA_TFLOPS, "bf16 MFMA dense (2500), 1 MFMA per product"),
    "p6": (PEAK_F16_MFMA_TFLOPS * 1.5, "bf16 MFMA dense (2500) x 3/2 (Winograd F(2,3) along depth issues "
                                       "2/3 MFMA per algorithmic product)"),
}


def tag_peak(tag, fallback):
    for k, v in PEAK_OF_TAG.items():
        if ("_%s_" % k) in tag:
            return v
    return fallback
# precision -> (dtype field, description of the conv arithmetic, basis of the roofline peak)
ARITH = {
    "f32": ("f32", "exact fp32 MFMA", "fp32 MFMA dense"),
    "f16x3": ("f32 (conv products: 3x f16-split MFMA, f32 accumulate)",
              "fp32 data/accumulators; each product of the convs = 3 f16 MFMAs on hi/lo-split "
              "operands (error below fp32 accumulation error)",
              "f16 MFMA dense (2500) / 3 MFMAs per algorithmic fp32 product"),
    "f16": ("f16",
            "conv operands rounded to f16 (one MFMA per product) and the residual stream stored in f16, as the "
            "reference's --use_fp16 torso (unet.py:1035); fp32 accumulation, GroupNorm statistics, timestep "
            "path, network input and output", "f16 MFMA dense"),
    "bf16": ("bf16",
             "conv operands rounded to bf16 (one bf16 MFMA per product) and the residual stream stored in "
             "bf16; fp32 accumulation, GroupNorm statistics, timestep path, network input and output",
             "bf16 MFMA dense"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(arch, respacing, device):
    from guided_diffusion import script_util as su
    from guided_diffusion import synth
    fl = su.sr_model_and_diffusion_defaults()
    fl.update(arch, timestep_respacing=respacing)
    model, diff = su.sr_create_model_and_diffusion(**fl)
    t0 = time.time()
    sd = {k: torch.from_numpy(synth.synth_param(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    model.to(device).eval()
    log("[bench] %d parameters synthesised in %.1fs" % (sum(v.numel() for v in sd.values()), time.time() - t0))
    return model, diff, sd


def host_cpu():
    """(model string, physical cores, cores this process may run on) of the host."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model == "unknown":
                    model = v
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not k and pid is not None:
                    phys.add((pid, cid))
                    pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    return model, (len(phys) or (os.cpu_count() or 1)), aff


def cpu_quota_cores():
    """CPU time this process's cgroup is granted, in cores (cgroup v2 cpu.max or v1 cfs quota / period), walking up
    from the process's own cgroup; None = no quota set.  The record of what the box grants, next to what it shows."""
    paths = []
    try:
        with open("/proc/self/cgroup") as f:
            for line in f:
                _, ctrl, rel = line.strip().split(":", 2)
                if ctrl == "":                                  # v2
                    d = "/sys/fs/cgroup" + rel
                    while True:
                        paths.append((os.path.join(d, "cpu.max"), None))
                        if d in ("/sys/fs/cgroup", "/"):
                            break
                        d = os.path.dirname(d)
                elif "cpu" in ctrl.split(","):                  # v1
                    d = "/sys/fs/cgroup/cpu" + rel
                    paths.append((os.path.join(d, "cpu.cfs_quota_us"), os.path.join(d, "cpu.cfs_period_us")))
    except (OSError, ValueError):
        pass
    paths += [("/sys/fs/cgroup/cpu.max", None),
              ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")]
    best = None
    for quota_file, period_file in paths:
        try:
            with open(quota_file) as f:
                txt = f.read().split()
            if period_file is None:
                if not txt or txt[0] == "max":
                    continue
                q, per = float(txt[0]), float(txt[1])
            else:
                q = float(txt[0])
                if q <= 0:
                    continue
                with open(period_file) as f:
                    per = float(f.read().split()[0])
            best = q / per if best is None else min(best, q / per)
        except (OSError, ValueError, IndexError):
            continue
    return best


def psnr_db(got, ref):
    """10 log10(range^2 / MSE), range = max - min of the reference volume."""
    mse = float(((got.double() - ref.double()) ** 2).mean())
    rng = float(ref.max() - ref.min())
    return float("inf") if mse == 0 else 10.0 * float(np.log10(rng * rng / mse))


def rel_err(got, ref):
    return float((got.double() - ref.double()).abs().max() / ref.double().abs().max())


def cpu_baseline(arch, sd, size, respacing, n_steps, threads):
    """Time the oracle's p_sample steps on the host cores (bounded sample).  Returns the record
    and the oracle's state after those steps (the checker of parity (b))."""
    from guided_diffusion import synth
    from oracle import sampler_ref, schedule_ref, unet_ref
    torch.set_num_threads(threads)
    cfg = unet_ref.sr_config(**arch)
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", respacing)
    T = len(tmap)
    shape = (1, 1, size, size, size)
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, n_steps + 2, seed=10)]
    img = draws[0]
    times = []
    with torch.no_grad():
        for k in range(n_steps + 1):   # first one is warm-up
            i = T - 1 - k
            t0 = time.time()
            out = unet_ref.unet_forward(sd, cfg, img, torch.full((1,), tmap[i], dtype=torch.long), lr)
            mean, log_var, _ = sampler_ref.mean_variance(tb, out, img, i, True, False, True)
            img = mean + torch.exp(0.5 * log_var) * draws[k + 1]
            times.append(time.time() - t0)
            log("[bench] cpu oracle step %d: %.2fs" % (k, times[-1]))
    per_step = float(np.mean(times[1:]))
    cpu_model, phys, aff = host_cpu()
    return {
        "value": 1.0 / (per_step * T),
        "unit": "volumes/s",
        "cores": threads,
        "cpu_model": cpu_model,
        "physical_cores": phys,
        "cores_available": aff,
        # what the box's cgroup actually grants this process (None: no quota): on the pool's one-GPU boxes every
        # host core is visible (cores_available) but the CPU-time share is this many cores
        "cpu_quota_cores": cpu_quota_cores(),
        "kind": "port",
        "sample": "%d p_sample steps (after 1 warm-up) of the same 1x%d^3 published-arch workload, "
                  "%.2f s/step, extrapolated x%d steps" % (n_steps, size, per_step, T),
    }, img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddpm-steps", type=int, default=250)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1, help="volumes per GPU per step")
    ap.add_argument("--arch", choices=["published", "tiny"], default="published")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--f32-steps", type=int, default=-1,
                    help="DDPM steps of the first timed volume repeated in the exact-fp32 arithmetic "
                         "(exact_f32 sub-record and parity (a)); -1 = all of them, 0 = skip; only "
                         "with the default f16x3 precision")
    ap.add_argument("--probe-ms", type=float, default=50.0,
                    help="length of the MFMA calibration loop (roofline.device_sustained_tflops; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for cpu_baseline")
    ap.add_argument("--golden", type=int, choices=[0, 1], default=1,
                    help="one more volume against the reference's own 250-step sample (tests/golden/sampler250_64.npz)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--precision", choices=["f32", "f16x3", "f16", "bf16"],
                    default=os.environ.get("DDPM3D_PRECISION", "f16x3"),
                    help="arithmetic of the conv products (all keep fp32 data and accumulators); "
                         "f16 = the reference's --use_fp16 analogue (BASELINE config 4)")
    ap.add_argument("--large-size", type=int, default=None, help="override the factory's large_size flag")
    ap.add_argument("--attention-resolutions", default=None,
                    help="override, e.g. '16' with --large-size 128 --size 128 = BASELINE config 5")
    ap.add_argument("--step-graph", type=int, choices=[0, 1], default=None,
                    help="replay a captured hipGraph per forward (1) or issue the launches one by one (0); "
                         "default: the model's own setting")
    ap.add_argument("--sampler", choices=["ddpm", "ddim"], default="ddpm",
                    help="ddim: --ddpm-steps DDIM steps (timestep_respacing ddimN, eta 0)")
    args = ap.parse_args()

    # `bench.py --gpus N` by itself: start the N ranks as a CHILD process (this parent has made no
    # GPU call and never makes one), relay rank 0's JSON line, exit with the child's status.
    if args.gpus > 1 and "RANK" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
        child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
        lines = [ln for ln in child.stdout.decode().splitlines() if ln.startswith("{")]
        if child.returncode == 0 and len(lines) != 1:
            log("[bench] expected one JSON line from rank 0, got %d" % len(lines))
            sys.exit(1)
        for ln in lines:
            print(ln, flush=True)
        sys.exit(child.returncode)

    # stdout carries exactly ONE JSON line.  Libraries (RCCL prints a version banner) write to
    # fd 1 directly, so fd 1 is pointed at stderr for the run and the result goes to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # checked before anything touches a GPU or a process group: a launcher whose world size is not --gpus
    # (e.g. BASELINE config 3 started with --gpus 8 under a 4-rank launcher) is refused by every rank at once
    if args.gpus != world:
        os.dup2(real_stdout, 1)
        raise SystemExit("[bench] --gpus %d but WORLD_SIZE %d: launch one rank per GPU (plain "
                         "`python bench.py --gpus N` starts them itself)" % (args.gpus, world))
    # --dist-backend gloo --share-gpu: rehearsal of the multi-rank flow on a one-GPU box (all
    # ranks on cuda:0, collectives through host memory).  The real runs use RCCL ("nccl").
    gloo = args.dist_backend == "gloo"
    dev_index = 0 if args.share_gpu else local
    # a process group also for one rank when launched under torch.distributed.run with
    # DDPM3D_BENCH_FORCE_DIST=1 (exercises the RCCL code path on a one-GPU box)
    use_dist = world > 1 or (os.environ.get("DDPM3D_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    collective_ranks, collective_backend = 1, None
    if use_dist:
        # the number of ranks the collective library itself sees: one all_gather of the rank ids
        ids = [torch.zeros(1, dtype=torch.int64, device=torch.device("cpu") if gloo else torch.device("cuda", dev_index))
               for _ in range(world)]
        dist.all_gather(ids, torch.full((1,), rank, dtype=torch.int64, device=ids[0].device))
        assert sorted(int(t.item()) for t in ids) == list(range(world))
        collective_ranks, collective_backend = dist.get_world_size(), dist.get_backend()
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    coll_dev = torch.device("cpu") if gloo else device

    from guided_diffusion import synth
    arch = dict(PUBLISHED if args.arch == "published" else TINY)
    if args.large_size is not None:
        arch.update(large_size=args.large_size, small_size=args.large_size)
    if args.attention_resolutions is not None:
        arch.update(attention_resolutions=args.attention_resolutions)
    # architecture overrides are named in config.workload (config 5 adds five attention blocks)
    overrides = ""
    if args.large_size is not None or args.attention_resolutions is not None:
        overrides = "; OVERRIDES large_size=%s attention_resolutions=%s" % (
            arch["large_size"], arch["attention_resolutions"])
    respacing = ("ddim%d" % args.ddpm_steps) if args.sampler == "ddim" else str(args.ddpm_steps)
    model, diff, sd = build_model(arch, respacing, device)
    model.conv_precision = args.precision
    if args.step_graph is not None:
        model.step_graph = bool(args.step_graph)
    # roofline peak for the dominant kernel's arithmetic: fp32 MFMA; f16 MFMA / 3 (three f16
    # MFMAs per algorithmic fp32 product); f16 MFMA (one per product)
    peak = {"f32": PEAK_F32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS / 3.0,
            "f16": PEAK_F16_MFMA_TFLOPS, "bf16": PEAK_F16_MFMA_TFLOPS}[args.precision]
    B, S = args.batch, args.size
    shape = (B, 1, S, S, S)
    lr = torch.from_numpy(np.stack([synth.synth_low_res((1, S, S, S), seed=1234 + rank * 1000 + b)
                                    for b in range(B)])).to(device)
    eng = model.engine()
    plan = eng.plan(B, S, S, S)
    flops_fwd = sum(f for _, f in plan.conv_meta.values())
    if overrides:
        n_attn = sum(1 for l in model.topology.all_layers() if l.kind == "attn")
        overrides += " (%d attention blocks)" % n_attn
    T = diff.num_timesteps

    loop = diff.ddim_sample_loop_progressive if args.sampler == "ddim" else diff.p_sample_loop_progressive

    def one_volume(step_index, measure, max_steps=None):
        # per-volume noise keyed by the GLOBAL volume index, so results do not depend on world size
        gen = torch.Generator(device=device)
        gen.manual_seed(10 + step_index * world + rank)
        noise = torch.randn(*shape, device=device, generator=gen)
        torch.manual_seed(1000 + step_index * world + rank)
        timing = []
        k = 0
        final = None
        for final in loop(model, shape, noise, model_kwargs={"low_res": lr}):
            k += 1
            plan.timing = timing if (measure and k == T // 2) else None   # instrument ONE forward
            if max_steps is not None and k >= max_steps:
                break
        plan.timing = None
        sample = final["sample"]
        if max_steps is not None:
            return sample, timing
        if use_dist:
            mine = sample.to(coll_dev)
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
        return sample, timing

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        t0 = time.time()
        one_volume(-1 - w, False)
        torch.cuda.synchronize()
        log("[bench] rank %d warm-up volume %d: %.2fs" % (rank, w, time.time() - t0))

    timings = []
    first_sample = None
    sync()
    t_start = time.time()
    for s in range(args.steps):
        sample, tm = one_volume(s, True)
        if s == 0:
            first_sample = sample.clone()
        timings += tm
        if rank == 0:
            torch.cuda.synchronize()
            log("[bench] step %d done at %.2fs" % (s, time.time() - t_start))
    sync()
    elapsed = time.time() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(sample).all()

    # roofline of the conv kernel family, from the instrumented forwards
    by_tag = {}
    for tag, fl, e0, e1 in timings:
        ms = e0.elapsed_time(e1)
        a = by_tag.setdefault(tag, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += fl
        a[2] += ms
    tot_fl = sum(a[1] for a in by_tag.values())
    tot_ms = sum(a[2] for a in by_tag.values())
    dom = max(by_tag.items(), key=lambda kv: kv[1][2]) if by_tag else None
    roof = None
    if dom:
        tag, (cnt, fl, ms) = dom
        ach = fl / (ms * 1e-3) / 1e12
        # HBM-side bytes per launch of the same kernel over the same launch mix: PMC counters
        # cannot be read from inside this process, so they come from the committed rocprofv3
        # pass over THIS workload (tools/pmc_bench.sh -> profiles/rNN_pmc_traffic.json);
        # null for any other configuration.
        traffic, traffic_src = None, None
        # (the last field names the dominant kernel's tile form: bump it with any change to that kernel's memory
        # behaviour, so that a traffic file measured on an older form stops matching instead of going stale)
        sig = "%s|%d|%d|%s|%s|%s" % (args.precision, S, B, args.arch, args.attention_resolutions, TRAFFIC_EPOCH)
        here = os.path.dirname(os.path.abspath(__file__))
        for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_traffic.json")), reverse=True):
            with open(f) as fh:
                tj = json.load(fh)
            k = tj.get("kernels", {}).get(tag)
            if tj.get("signature") == sig and k:
                traffic, traffic_src = round(k["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(f)
                break
        kpeak, kbasis = tag_peak(tag, (peak, ARITH[args.precision][2]))
        roof = {
            "bound": "mfma", "kernel": tag, "achieved": round(ach, 2), "peak": round(kpeak, 1),
            "unit": "TFLOP/s", "frac": round(ach / kpeak, 4), "traffic": traffic,
            "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
            # PMC counters cannot be collected inside this process: the figure is the committed
            # rocprofv3 pass over this same workload, not a measurement of this run
            "traffic_static": traffic is not None,
            "traffic_signature": sig,
            "peak_basis": kbasis,
            "launches_timed": cnt, "avg_launch_ms": round(ms / cnt, 4),
            "all_conv_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                 "ms_per_forward": round(tot_ms / max(1, args.steps), 3),
                                 "gflop_per_forward": round(flops_fwd / 1e9, 1)},
            "per_kernel": {t: {"launches": a[0], "tflops": round(a[1] / (a[2] * 1e-3) / 1e12, 2),
                               "ms": round(a[2] / max(1, args.steps), 3)} for t, a in sorted(by_tag.items())},
        }

    # MFMA calibration of this board (outside the timed region): the instruction the dominant kernel
    # issues, two waves per SIMD like that kernel, register-only, ~50 ms
    from guided_diffusion import _hip as HH
    sustained = None
    if roof is not None and args.probe_ms > 0 and rank == 0:
        lib = HH.load()
        kind = {"f32": HH.PROBE_F32_32X32X2, "bf16": HH.PROBE_BF16_32X32X16}.get(args.precision, HH.PROBE_F16_32X32X16)
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        blocks = 2 * cus
        pout = torch.empty(blocks * 256, dtype=torch.float32, device=device)
        pclk = torch.zeros(blocks * 2, dtype=torch.int64, device=device)
        st = HH.stream()
        fl_iter = lib.ddpm3d_mfma_probe_flops_per_iter(kind)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters, ms = 4000, 0.0
        for attempt in range(3):   # warm launch, sizing launch, the timed one
            e0.record()
            HH.check(lib.ddpm3d_mfma_probe(kind, iters, blocks, HH.ptr(pout), HH.ptr(pclk), st))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            if attempt == 1:
                iters = max(1000, int(iters * args.probe_ms / max(ms, 1e-3)))
        c = pclk.view(blocks, 2).double().cpu()
        sustained = {"tflops": fl_iter * iters * blocks / (ms * 1e-3) / 1e12,
                     "ghz": float((c[:, 0] / c[:, 1]).median()) * 0.1, "ms": ms}
        mfma_per_product = (PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS) / roof["peak"]
        roof["device_sustained_tflops"] = round(sustained["tflops"], 1)
        roof["device_sustained_clock_ghz"] = round(sustained["ghz"], 3)
        roof["device_sustained_basis"] = ("ddpm3d_mfma_probe: register-only loop of the dominant kernel's MFMA "
                                          "instruction, 2 waves/SIMD on every CU, pseudo-random operands, "
                                          "%.0f ms, run after the timed region" % ms)
        roof["frac_of_sustained"] = round(roof["achieved"] * mfma_per_product / sustained["tflops"], 4)
        log("[bench] device sustains %.0f TFLOP/s of MFMA issue at %.2f GHz; dominant kernel at %.3f of it"
            % (sustained["tflops"], sustained["ghz"], roof["frac_of_sustained"]))

    # The first timed volume again (same noise, same per-step draws) in the EXACT fp32 arithmetic
    # (v_mfma_f32_32x32x2_f32): the driver-timed record then carries an exact-arithmetic number beside
    # the default one, and -- over all T steps -- the parity of the default arithmetic against it.
    exact = None
    parity = {}
    if args.precision == "f16x3" and args.f32_steps != 0 and rank == 0 and world == 1:
        model.conv_precision = "f32"
        eng32 = model.engine()
        k_steps = T if args.f32_steps < 0 else min(args.f32_steps, T)
        plan_default, plan = plan, eng32.plan(B, S, S, S)
        one_volume(0, False, max_steps=2)          # warm-up: plan, weight packing
        torch.cuda.synchronize()
        t0 = time.time()
        sample32, _ = one_volume(0, False, max_steps=k_steps)
        torch.cuda.synchronize()
        dt = time.time() - t0
        fl32 = sum(f for _, f in plan.conv_meta.values())
        plan = plan_default
        tf = fl32 * k_steps / dt / 1e12
        exact = {"precision": "f32 (v_mfma_f32_32x32x2_f32, exact fp32 products)", "steps_timed": k_steps,
                 "ms_per_ddpm_step": round(1000.0 * dt / k_steps, 3),
                 "value": B / (dt / k_steps * T), "unit": "volumes/s" + ("" if k_steps == T else
                                                                        " (extrapolated to %d steps)" % T),
                 "tflops": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                 "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                 "note": "whole step (all kernels), wall clock; conv FLOPs only"}
        model.conv_precision = args.precision
        log("[bench] exact fp32 arithmetic: %d steps in %.2fs (%.1f TFLOP/s)" % (k_steps, dt, tf))
        if k_steps == T and first_sample is not None:
            parity["vs_exact_f32"] = {
                "what": "first timed volume (all %d steps) vs the same volume, same noise, in this library's "
                        "exact-fp32 arithmetic" % T,
                "rel_err": rel_err(first_sample, sample32), "psnr_db": round(psnr_db(first_sample, sample32), 2)}
            log("[bench] parity vs exact f32: rel %.2e, PSNR %.1f dB"
                % (parity["vs_exact_f32"]["rel_err"], parity["vs_exact_f32"]["psnr_db"]))

    # "PSNR vs ref" of the metric, on the headline workload itself: ONE more volume of the same configuration with
    # the weights, conditioning volume and injected noise of tests/golden/sampler250_64.npz -- the REFERENCE's own
    # 250-step run of BASELINE config 2 (generated in the build container, tests/golden/make_golden.py) -- against
    # that file's final sample.  A fixture, not the oracle: nothing under /root/reference is read here.
    gold = os.path.join(ROOT, "tests", "golden", "sampler250_64.npz")
    if (rank == 0 and world == 1 and args.arch == "published" and S == 64 and T == 250 and args.sampler == "ddpm"
            and not overrides and os.path.exists(gold) and args.golden):
        from guided_diffusion import synth as _synth
        g = np.load(gold)
        shp = (1, 1, S, S, S)
        draws = [torch.from_numpy(a).to(device) for a in _synth.synth_noise(shp, T + 1, seed=10)]
        lrg = torch.from_numpy(_synth.synth_low_res(shp, seed=1234)).to(device)
        got = diff.p_sample_loop(model, shp, draws[0], model_kwargs={"low_res": lrg}, step_noise=draws[1:]).cpu()
        ref = torch.from_numpy(g["sample"])
        parity["vs_reference"] = {
            "what": "this configuration (1x64^3, %d DDPM steps, published architecture, %s arithmetic) on the weights, "
                    "conditioning volume and injected noise of tests/golden/sampler250_64.npz, against that file: the "
                    "reference's own final sample" % (T, args.precision),
            "rel_err": rel_err(got, ref), "psnr_db": round(psnr_db(got, ref), 2)}
        del draws
        log("[bench] parity vs the reference's own 250-step run: rel %.2e, PSNR %.1f dB"
            % (parity["vs_reference"]["rel_err"], parity["vs_reference"]["psnr_db"]))

    if rank == 0:
        vols = args.steps * B * world
        res = {
            # BASELINE.json's metric string, verbatim, for the configuration it is quoted on
            "metric": ("denoised 64³ volumes/sec @250 DDPM steps, 1/2/4/8 MI355X; PSNR vs ref"
                       if args.sampler == "ddpm" and T == 250 and S == 64
                       else "denoised %d^3 volumes/sec @%d %s steps" % (S, T, args.sampler.upper())),
            "value": vols / elapsed,
            "unit": "volumes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # fp32 tensors and accumulators in both modes; f16x3 = every fp32 product of the 3x3x3
            # convs formed as three f16 MFMAs on hi/lo-split operands (fp32-grade error, see tests)
            "dtype": ARITH[args.precision][0],
            "data": "synthetic",
            "config": {"workload": "%dx1x%d^3 volume(s) per GPU, %d %s steps, %s architecture "
                                   "(SuperResModel_noatt, %d base ch, mult (1,1,2,3,4), %d res blocks, "
                                   "learn_sigma%s), seeded random weights, device RNG"
                                   % (B, S, T, args.sampler.upper(), args.arch, arch["num_channels"],
                                      arch["num_res_blocks"], overrides),
                       "parallelism": "independent volumes per rank (dp%d), all_gather of finished samples" % world,
                       "tflop_per_volume": round(flops_fwd * T / B / 1e12, 1),
                       "conv_arithmetic": ARITH[args.precision][1],
                       "launch": "hipGraph replay per forward" if model.step_graph else "one launch per kernel from the host"},
            # ranks the collective library saw in an all_gather, and which library: "nccl" = RCCL over xGMI;
            # "gloo" = the one-GPU rehearsal through host memory.  rccl_ranks is set for RCCL runs only.
            "collective_ranks": collective_ranks,
            "collective_backend": collective_backend,
            "rccl_ranks": collective_ranks if collective_backend == "nccl" else None,
            "roofline": roof,
            "exact_f32": exact,
            "parity": parity or None,
        }
        if world == 1 and args.cpu_steps > 0:
            threads = os.cpu_count() or 1
            try:
                threads = len(os.sched_getaffinity(0))
            except Exception:
                pass
            # threads = min(visible cores, the cgroup's CPU quota if one is set, --cpu-threads): a one-GPU box
            # of the pool shows every host core and grants a share of them (the record carries both; thread
            # sweep: tools/cpu_threads_sweep.py -> profiles/r04_cpu_threads_sweep.txt)
            quota = cpu_quota_cores()
            if quota is not None:
                threads = max(1, min(threads, int(quota + 0.5)))
            threads = min(threads, args.cpu_threads)
            sd_cpu = {k: v for k, v in sd.items()}
            res["cpu_baseline"], cpu_img = cpu_baseline(arch, sd_cpu, S, respacing, args.cpu_steps, threads)
            # parity (b): the same first reverse steps on the GPU, the oracle's noise injected
            n_or = args.cpu_steps + 1
            draws = [torch.from_numpy(a).to(device) for a in synth.synth_noise((1, 1, S, S, S), n_or + 1, seed=10)]
            lr1 = torch.from_numpy(synth.synth_low_res((1, 1, S, S, S), seed=1234)).to(device)
            fin = None
            for k, fin in enumerate(diff.p_sample_loop_progressive(model, (1, 1, S, S, S), draws[0],
                                                                   model_kwargs={"low_res": lr1},
                                                                   step_noise=draws[1:] + [draws[1]] * T)):
                if k + 1 >= n_or:
                    break
            got = fin["sample"].cpu()
            res["parity"] = dict(res["parity"] or {})
            res["parity"]["vs_cpu_oracle"] = {
                "what": "state after the first %d reverse steps (injected noise) vs oracle/ on the host, "
                        "which is pinned to the reference's own outputs (tests/golden)" % n_or,
                "rel_err": rel_err(got, cpu_img), "psnr_db": round(psnr_db(got, cpu_img), 2)}
            log("[bench] parity vs CPU oracle after %d steps: rel %.2e, PSNR %.1f dB"
                % (n_or, res["parity"]["vs_cpu_oracle"]["rel_err"], res["parity"]["vs_cpu_oracle"]["psnr_db"]))
        else:
            res["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
