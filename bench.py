#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config {2,4}] [--no-cpu-baseline]

A "step" is one pass of FlowHighSR.generate() over one batch of synthetic clips whose low-rate
input and prior noise are already resident in HBM (device resampler -> log-mel -> `time_step`
vector-field evaluations -> BigVGAN -> STFT post-processing), ending with the 48 kHz waveform in HBM.

--config 2 (default): BASELINE.json configs[1] on every GPU (B = 1, one 10 s clip, 12 -> 48 kHz, time_step = 1
    euler, transformer 2 x 16 x 64, SYNTH-CFG BigVGAN-48k-256band; random-init weights, synthetic audio).
    For N > 1 every rank runs that workload on its own clips (independent clips, no data-path collective:
    weak scaling); after the timed region the clips of all ranks go once through the RCCL scatter / gather path
    (flowhigh_amd.parallel.generate_sharded) and rank 0 checks the result against its own runs.
--config 4: BASELINE.json configs[3]: 32 N clips of 10 s, 8 -> 48 kHz, euler x 1, live on rank 0; a step is
    scatter (RCCL P2P over xGMI) -> generate on every rank -> gather on rank 0 (256 clips over 8 GPUs).

`--gpus N` with N > 1 works both ways: under torch.distributed.run (RANK / WORLD_SIZE in the environment: one
rank per GPU), or invoked directly, in which case this script starts that launcher as a child process BEFORE
anything touches a GPU and exits with its code.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
SECS, STEPS_ODE, METHOD = 10.0, 1, "euler"
CONFIGS = {2: dict(sr_in=12000, per_gpu=1, sharded=False, name="BASELINE configs[1]"),
           4: dict(sr_in=8000, per_gpu=32, sharded=True, name="BASELINE configs[3]")}
EVENT_EVERY = 8          # HIP events bracket the conv / activation launches of every 8th timed step


def cpu_baseline(sd, cfg, sr_in):
    """The oracle (CPU restatement, kind 'port') on the metric's own unit of work: ONE 10 s clip of the workload,
    same weights and path, one warm-up run, then the median of 3 with the spread stated:
    ~20 s per run on the GPU box's host, ~90 s in all."""
    from flowhigh_amd import synth
    from oracle import ref_cpu
    audio = synth.lowres_clip(0, SECS, sr_in)
    noise = synth.prior_noise(0, int(SECS * 100))
    # torch's intra-op pool degrades badly past a few dozen threads on these small convs (256 threads
    # on the GPU box's host: 50x slower than 16), so the baseline uses at most 16 threads and says so.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    ref_cpu.generate(sd, cfg, audio, sr_in, noise, STEPS_ODE, METHOD)
    warm = time.perf_counter() - t0
    runs = 3
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        ref_cpu.generate(sd, cfg, audio, sr_in, noise, STEPS_ODE, METHOD)
        times.append(time.perf_counter() - t0)
    return {"value": round(SECS / statistics.median(times), 4), "unit": "audio-seconds/s", "cores": threads,
            "kind": "port",
            "runs_s": [round(t, 2) for t in times],
            "sample": f"one {SECS:g} s clip = one batch-1 step of this workload, same weights and path, 1 warm-up "
                      f"({warm:.1f} s) + median of {runs} runs ({min(times):.1f} .. {max(times):.1f} s), {threads} torch "
                      f"threads of {avail} visible host CPUs"}


DTYPE_BY_FORM = {
    "winograd": "f32",
    "direct": "f32",
    "bf16x6": "f32 in / out / accumulate; the products of the residual-stack convs (Winograd wide stages, direct narrow stages) and of "
              "the transformer's linears as 6 bf16 MFMAs over exact 3-piece splits (conv_form='bf16x6': fp32-grade, dropped terms "
              "<= 2^-24 |a b|); everything else fp32 arithmetic",
}
KERNEL_BY_FAMILY = {
    "wino54": "conv_wino54_kernel (Winograd F(5,4) wide-stage conv, v_mfma_f32_32x32x2_f32)",
    "wino54_bf16x6": "conv_wino54_kernel<BF> (Winograd F(5,4) wide-stage conv, 6 x v_mfma_f32_32x32x16_bf16 per fp32 k-block)",
    "wino43": "conv_wino_kernel (Winograd F(4,3): conv_pre, first two upsamplers, v_mfma_f32_32x32x2_f32)",
    "wino43_bf16x6": "conv_wino_kernel<BF> (Winograd F(4,3): conv_pre, first four upsamplers, bf16 x 6)",
    "amp": "amp_actconv_kernel (narrow-stage Winograd F(5,4) conv, v_mfma_f32_16x16x4_f32)",
    "narrow_bf16x6": "narrow_bf_kernel (narrow-stage direct conv, 6 x v_mfma_f32_16x16x32_bf16 per fp32 k-block)",
    "direct": "conv_mfma_kernel (direct implicit-GEMM conv: the upsamplers below 768 (bf16 x 6 form: 192) input channels, v_mfma_f32_32x32x2_f32)",
}
PEAK_BY_FAMILY = {"wino54_bf16x6": 2500.0, "wino43_bf16x6": 2500.0, "narrow_bf16x6": 2500.0}      # dense bf16 MFMA; every other family: the fp32 MFMA peak


def family_split(conv_ev, launches, timed_steps):
    """roofline.by_family: the conv launches of the sampled steps by kernel family (the plan tags every launch: planner.conv_launches),
    each with launches per step, ms per step, executed GFLOP per step, TFLOP/s on its own matrix instructions and the fraction of
    THAT pipe's dense peak (bf16 x 6 families: 6 bf16 MFMA FLOPs per executed fp32-equivalent FLOP against 2.5 PFLOP/s)."""
    fam = {}
    n = len(launches)
    for i, (a, b) in enumerate(conv_ev):
        name, ex, alg = launches[i % n]
        f = fam.setdefault(name, dict(launches=0, ms=0.0, executed=0.0, algorithmic=0.0))
        f["launches"] += 1
        f["ms"] += a.elapsed_time(b)
        f["executed"] += ex
        f["algorithmic"] += alg
    out = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        bf = name.endswith("_bf16x6")
        issued = f["executed"] * (6.0 if bf else 1.0)
        peak = PEAK_BY_FAMILY.get(name, PEAK_FP32_MFMA_TFLOPS)
        tf = issued / (f["ms"] / 1e3) / 1e12 if f["ms"] > 0 else 0.0
        out[name] = {"launches_per_step": f["launches"] // max(timed_steps, 1), "ms_per_step": round(f["ms"] / max(timed_steps, 1), 3),
                     "executed_gflop_per_step": round(f["executed"] / max(timed_steps, 1) / 1e9, 1),
                     "algorithmic_gflop_per_step": round(f["algorithmic"] / max(timed_steps, 1) / 1e9, 1),
                     "matrix_tflops": round(tf, 1), "matrix_instructions": "bf16 (6 per fp32 product)" if bf else "fp32",
                     "peak": peak, "frac": round(tf / peak, 4)}
    return out


def alt_form(form, sd, cfg, dev, sr_in, x, z, out_main, B, n_frames, steps):
    """NOT the headline: the same workload with the vocoder's convs in another arithmetic form (FLowHigh(..., conv_form=form)),
    reported next to the headline with its distance from the headline form's waveform."""
    from flowhigh_amd import FLowHigh, FlowHighSR
    model = FlowHighSR(FLowHigh(sd, cfg, dev, conv_form=form), torchdiffeq_ode_method=METHOD, upsampling_method="hip")
    voc = model.flowhigh.vocoder
    for _ in range(3):
        out = model.generate_from_device(x, sr_in, STEPS_ODE, noise=z)
    conv_ev, timed = [], 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        voc.conv_timing = conv_ev if i % EVENT_EVERY == 0 else None
        timed += i % EVENT_EVERY == 0
        out = model.generate_from_device(x, sr_in, STEPS_ODE, noise=z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    voc.conv_timing = None
    conv_ms = sum(a.elapsed_time(b) for a, b in conv_ev) / max(timed, 1)
    plan = voc.plan(B, n_frames)
    return {"conv_form": form, "value": round(B * SECS / dt, 3), "unit": "audio-seconds/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "conv_ms_per_step": round(conv_ms, 3), "act_blocks_per_cu": voc.act_blocks, "dtype": DTYPE_BY_FORM[form],
            "max_abs_diff_vs_headline_waveform": float((out - out_main).abs().max().item()),
            "by_family": family_split(conv_ev, plan["conv_launches"], timed),
            "note": f"the same workload with conv_form='{form}' (FlowHighSR.from_local(..., conv_form=); INTEGRATION.md section 1); not the headline"}


def visible_gpus():
    """GPUs of this node counted WITHOUT the HIP runtime: the kfd topology nodes that have SIMDs (CPUs have none),
    cut by *_VISIBLE_DEVICES.  None when the topology is not readable (the ranks then find out themselves)."""
    if not Path("/sys/class/kfd").exists():       # no amdgpu compute driver: no AMD GPU
        return 0
    try:
        nodes = list(Path("/sys/class/kfd/kfd/topology/nodes").iterdir())
        n = 0
        for node in nodes:
            props = dict(line.split()[:2] for line in (node / "properties").read_text().splitlines() if line.strip())
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([v for v in os.environ[var].split(",") if v.strip()]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (never exec:
    nothing here has touched a GPU yet, and nothing will in this process)."""
    # (the parent never asks the HIP runtime anything, not even the device count: without amdsmi torch falls back to
    # hipGetDeviceCount, which initialises the runtime; a rank whose device is missing fails loudly by itself)
    n_dev = visible_gpus()
    if n_dev is not None and n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 1 for --config 2, 32 for --config 4)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the side measurement of the other conv form (fp32-MFMA Winograd next to a bf16 x 6 headline, or the reverse)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("FH_BENCH_GRAPH", "0")),
                    help="1: a step replays the HIP graph of generate_from_device (same launches, one enqueue)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:           # under a launcher, also with ONE rank: the RCCL init path runs on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)

    from flowhigh_amd import FLowHigh, FlowHighSR, parallel, synth
    conf = CONFIGS[args.config]
    sr_in = conf["sr_in"]
    cfg = synth.SYNTH_CFG
    sd = synth.make_state_dict(cfg, 0)
    # The model is loaded the way a deployment loads it: rank 0 packs the weights once into the flat blob
    # (flowhigh_amd/weights.py; on the CPU: float64 Winograd transforms of 118 M parameters), every rank maps that ONE file
    # and uploads it with one copy.  pack_s / load_s go into the line's config.
    from flowhigh_amd import convert, weights
    from flowhigh_amd.planner import resolve_conv_form
    # the arithmetic form of the vocoder's convs: what a deployment gets by default (conv_form='auto' resolves to it; FH_CONV_FORM
    # overrides).  No load-time probe here: the blob path has no checkpoint at hand, and the bench names its form in the line.
    form = resolve_conv_form()[0]
    blob = Path(os.environ.get("FH_BENCH_BLOB", f"/tmp/flowhigh_amd_bench_{os.getuid()}.blob"))
    pack_s = None
    if rank == 0:
        t0 = time.perf_counter()
        store = convert.build_store(sd, cfg, conv_form=form)
        store.save(blob, cfg, weights.format_tag(form), {})
        del store
        pack_s = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        # one activation-occupancy setting for the node, measured by rank 0 while the others wait (an explicit collective:
        # the model constructors never communicate)
        from flowhigh_amd import vocoder as _voc
        _voc.sync_act_blocks(dev, bf=form == "bf16x6")
    t0 = time.perf_counter()
    store = weights.WeightStore.open(blob, dev, expect_format=weights.format_tag(form))
    if store is None:
        raise SystemExit(f"bench.py: weight blob {blob} not usable: {weights.WeightStore.why}")
    model = FlowHighSR(FLowHigh(None, cfg, dev, store=store, conv_form=form), torchdiffeq_ode_method=METHOD, upsampling_method="hip")
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    blob_mib = blob.stat().st_size / 2 ** 20
    act_blocks = model.flowhigh.vocoder.act_blocks
    from flowhigh_amd import vocoder as _vocmod
    act_table = _vocmod.calibrate_act_occupancy.last_measurement        # [{cap: pair us}] per pass, None when the setting was given
    B = args.batch if args.batch is not None else conf["per_gpu"]
    n_frames = int(SECS * 100)
    n_in, t48 = int(SECS * sr_in), int(SECS * 48000)

    def make_inputs(indices):
        x = torch.stack([torch.from_numpy(synth.lowres_clip(i, SECS, sr_in)) for i in indices])
        z = torch.cat([synth.prior_noise(i, n_frames) for i in indices], 0)
        return x.to(dev), z.to(dev).contiguous()

    def gen(xs, ns):
        return model.generate_from_device(xs, sr_in, STEPS_ODE, noise=ns)

    if conf["sharded"]:
        # every clip of the job lives on rank 0; a step scatters them, runs them, gathers the waveforms
        x_all, z_all = make_inputs(range(world * B)) if rank == 0 else (None, None)

        def step():
            return parallel.generate_sharded(gen, x_all, z_all, n_in, n_frames, device=dev, n_total=world * B, t48=t48)
    else:
        x, z = make_inputs(range(rank * B, rank * B + B))

        def step():
            return gen(x, z)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    eager_step = step
    if args.graph and not conf["sharded"]:
        graphed = model.capture(B, n_in, sr_in, STEPS_ODE)       # the ~150 launches of a step recorded once
        graphed.x.copy_(x)
        graphed.noise.copy_(z.reshape(graphed.noise.shape))

        def step():                                               # noqa: F811
            return graphed.replay()
        for _ in range(2):
            out = step()
    voc = model.flowhigh.vocoder
    # HIP events (on the launch stream = torch's current stream) around every conv and every Activation1d launch
    # of every 8th timed step: an event pair costs ~6 us of stream time, sampling keeps that under 0.5 %
    conv_ev, act_ev, timed_steps = [], [], 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = i % EVENT_EVERY == 0
        voc.conv_timing = conv_ev if sampled else None
        voc.act_timing = act_ev if sampled else None
        timed_steps += sampled
        out = eager_step() if sampled else step()          # (HIP events cannot be recorded inside a graph replay)
    barrier()
    elapsed = time.perf_counter() - t0
    voc.conv_timing = voc.act_timing = None
    if not conf["sharded"] or rank == 0:
        rows = world * B if conf["sharded"] else B
        assert tuple(out.shape) == (rows, t48) and bool(torch.isfinite(out).all())
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernels ---------------------------------------------------------------------------
    plan = voc.plan(B, n_frames)
    conv_ms = sum(a.elapsed_time(b) for a, b in conv_ev)          # conv launches of the sampled timed steps
    act_ms = sum(a.elapsed_time(b) for a, b in act_ev)
    n_conv, n_act = len(conv_ev), len(act_ev)
    conv_s, act_s = conv_ms / 1e3 / max(n_conv, 1), act_ms / 1e3 / max(n_act, 1)
    # FLOPs the matrix cores execute per launch (Winograd F(5,4): 1.6 ceil(k/4), F(4,3): 1.5 ceil(k/3) instead of k MACs per output) and the
    # direct-form ("algorithmic") FLOPs of the same convs (2 622.6 MFLOP per frame)
    exec_per_launch = plan["conv_executed_flops"] * timed_steps / max(n_conv, 1)
    alg_per_launch = voc.conv_flops_per_frame() * n_frames * B * timed_steps / max(n_conv, 1)
    executed = exec_per_launch / conv_s / 1e12 if conv_s > 0 else 0.0
    alg_equiv = alg_per_launch / conv_s / 1e12 if conv_s > 0 else 0.0
    act_bytes_per_launch = plan["act_bytes"] * timed_steps / max(n_act, 1)
    act_gbs = act_bytes_per_launch / act_s / 1e9 if act_s > 0 else 0.0

    fams = family_split(conv_ev, plan["conv_launches"], timed_steps)
    dom = next(iter(fams)) if fams else None
    line = None
    if rank == 0:
        from tools.make_traffic_json import source_fingerprint
        fp_now = source_fingerprint(ROOT)
        stale = {}

        def pmc(name):          # (the PMC passes are per batch size: profiles/<name>.json at B = 1, <name>_B<n>.json otherwise)
            f = ROOT / "profiles" / (name if B == 1 else name.replace(".json", f"_B{B}.json"))
            if not f.exists():
                stale[name] = "no PMC file for this batch size"
                return None
            rec = json.loads(f.read_text())
            if rec.get("source_fingerprint") != fp_now:
                # the kernels or the launch planner changed since the PMC passes: the committed bytes describe other launches
                stale[name] = (f"profiles/{f.name} was measured on other kernel / planner sources (fingerprint "
                               f"{rec.get('source_fingerprint')}, now {fp_now}): re-run tools/profile_round.sh")
                return None
            return rec.get("bytes_per_launch")
        metric = "48 kHz audio-seconds/sec (real-time factor), 12→48 kHz, 10 s clips, 1/2/4/8 MI355X"
        if args.config == 4:
            metric = "48 kHz audio-seconds/sec (real-time factor), 8→48 kHz, 10 s clips, clips scattered from / gathered on rank 0"
        n_clips = world * B
        line = {
            "metric": metric,
            "value": round(n_clips * SECS * args.steps / elapsed, 3),
            "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_BY_FORM[form], "data": "synthetic", "hip_graph": bool(args.graph and not conf["sharded"]),
            "config": {"workload": f"{conf['name']}: B={B} per GPU x 10 s clip, {sr_in // 1000}->48 kHz, time_step=1 euler, "
                                   "transformer 2x16x64, BigVGAN-48k-256band SYNTH-CFG (rates 5,4,3,2,2,2; C0 1536), "
                                   "random-init weights",
                       "clips_per_gpu": B, "frames_per_clip": n_frames,
                       "parallelism": (f"{n_clips} clips on rank 0, RCCL P2P scatter -> generate -> gather, x{world}"
                                       if conf["sharded"] else f"clip-sharded x{world}, no data-path collective"),
                       "rccl_world_size": world if dist is not None else None,
                       "conv_form": form,
                       "act_blocks_per_cu": act_blocks,       # 0 = no cap; vocoder.calibrate_act_occupancy (same bits either way)
                       # what the calibration measured at load: per pass {cap: us of an (activation, conv) launch pair}
                       "act_calibration_pair_us": act_table,
                       # model construction from the packed weight blob (map + one H2D copy + the activation-occupancy
                       # calibration launches), and what packing it from the state dict took on this host (once, rank 0)
                       "load_s": round(load_s, 3), "pack_s": round(pack_s, 2), "blob_mib": round(blob_mib, 1),
                       "sharded_check": None},
            "roofline": {"bound": "mfma",
                         # the dominant kernel family of the step (most conv time; by_family has all of them)
                         "kernel": KERNEL_BY_FAMILY[dom],
                         "achieved": fams[dom]["matrix_tflops"], "peak": fams[dom]["peak"], "unit": "TFLOP/s",
                         "frac": fams[dom]["frac"],
                         "launches_per_step": fams[dom]["launches_per_step"],
                         "avg_launch_us": round(fams[dom]["ms_per_step"] * 1e3 / max(fams[dom]["launches_per_step"], 1), 2),
                         "traffic": pmc("conv_hbm_bytes_per_launch.json"),
                         "traffic_source": "profiles/conv_hbm_bytes_per_launch.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                           "passes of an earlier run of this command (tools/profile_round.sh), NOT measured "
                                           f"by this run; bytes per conv launch (all families) at B = {B}; null when that file was "
                                           "measured on other kernel / planner sources (traffic_stale says so)",
                         "note": "achieved = FLOPs the dominant family issues to ITS matrix instructions (Winograd F(5,4): 1.6 ceil(k/4) MACs per "
                                 "output and channel pair instead of k; bf16 x 6: six bf16 MFMA FLOPs per such fp32-equivalent FLOP, against the "
                                 "dense bf16 peak) / HIP-event time of that family's launches, measured live in this run; by_family: every "
                                 "conv kernel family the same way; all_conv: every conv launch together in fp32-equivalent executed FLOPs "
                                 "against the fp32 MFMA peak (the line's `frac` until round 5), algorithmic_equiv = direct-form FLOPs of the "
                                 "same convs (SURVEY.md 8d) / the same time",
                         "by_family": fams,
                         "all_conv": {"executed_fp32_equiv_tflops": round(executed, 2), "peak_fp32_mfma": PEAK_FP32_MFMA_TFLOPS,
                                      "frac_of_fp32_mfma_peak": round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "algorithmic_equiv": round(alg_equiv, 2),
                                      "algorithmic_equiv_frac": round(alg_equiv / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "launches_per_step": n_conv // max(timed_steps, 1),
                                      "avg_launch_us": round(conv_s * 1e6, 2),
                                      "executed_gflop_per_launch": round(exec_per_launch / 1e9, 3),
                                      "algorithmic_gflop_per_launch": round(alg_per_launch / 1e9, 3)},
                         "conv_ms_per_step": round(conv_ms / max(timed_steps, 1), 3)},
            "roofline_hbm": {"bound": "hbm",
                             "kernel": "act1d_strip_kernel (all Activation1d launches: up2x -> Snake(Beta) -> down2x)",
                             "achieved": round(act_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(act_gbs / PEAK_HBM_GBS, 4),
                             "traffic": pmc("act_hbm_bytes_per_launch.json"),
                             "traffic_source": "profiles/act_hbm_bytes_per_launch.json (rocprofv3 --pmc passes of an earlier "
                                               "run, as above)",
                             "note": "achieved = algorithmic bytes (each site reads and writes its [B, C, L] tensor once: "
                                     "8 B per sample) / HIP-event time of the activation launches",
                             "launches_per_step": n_act // max(timed_steps, 1),
                             "avg_launch_us": round(act_s * 1e6, 2),
                             "algorithmic_mb_per_launch": round(act_bytes_per_launch / 1e6, 2),
                             "act_ms_per_step": round(act_ms / max(timed_steps, 1), 3)},
        }
        for key, name in (("roofline", "conv_hbm_bytes_per_launch.json"), ("roofline_hbm", "act_hbm_bytes_per_launch.json")):
            if name in stale:
                line[key]["traffic_stale"] = stale[name]
        if not args.no_alt and world == 1 and args.config == 2:
            other = "winograd" if form == "bf16x6" else "bf16x6"
            line["alt_conv_form"] = alt_form(other, sd, cfg, dev, sr_in, x, z, out, B, n_frames, max(10, args.steps // 2))
        if not args.no_cpu_baseline and world == 1 and args.config == 2:
            line["cpu_baseline"] = cpu_baseline(sd, cfg, sr_in)
        else:
            line["cpu_baseline"] = None
    # ---- the scatter / gather path against rank-0-only runs (outside the timed region, after the line is built) ------
    # A watchdog on every rank: should the P2P exchange not finish, rank 0 still prints the measured line (without
    # the check) and every rank exits NON-ZERO, instead of leaving the launcher waiting or reporting success.
    rc = 0
    if dist is not None:
        import threading
        printed = threading.Lock()           # exactly one JSON line, whoever gets there first

        def emit():
            if rank == 0 and printed.acquire(blocking=False):
                print(json.dumps(line), flush=True)

        def bail():
            if rank == 0:
                line["config"]["sharded_check"] = "failed: not finished within 180 s: line printed without it"
            emit()
            os._exit(3)
        watchdog = threading.Timer(180.0, bail)
        watchdog.daemon = True
        watchdog.start()
        sharded_check = None
        try:
            if conf["sharded"]:
                xa, za, got = x_all, z_all, out
            else:
                xa, za = make_inputs(range(world * B)) if rank == 0 else (None, None)
                got = parallel.generate_sharded(gen, xa, za, n_in, n_frames, device=dev, n_total=world * B, t48=t48)
            if rank == 0:
                got = got.clone()
                same = all(torch.equal(gen(xa[s:s + B], za[s:s + B]), got[s:s + B]) for s in range(0, world * B, B))
                sharded_check = (f"{world * B} clips over {world} ranks through RCCL scatter/gather: "
                                 + ("bit-identical to rank-0-only runs" if same else "MISMATCH against rank-0-only runs"))
            torch.cuda.synchronize()
        except Exception as e:                   # noqa: BLE001  (report, keep the measured line, fail the run)
            sharded_check = f"failed: {type(e).__name__}: {e}"
        # rank 0 judges the comparison; any rank whose part of the exchange raised fails the run as well
        failed = (sharded_check is not None and sharded_check.startswith("failed")) or \
                 (rank == 0 and not (sharded_check and "bit-identical" in sharded_check))
        flag = torch.tensor([1 if failed else 0], device=dev)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)      # every rank leaves with rank 0's verdict
            rc = 4 if int(flag.item()) else 0
        except Exception:                        # noqa: BLE001
            rc = 4
        watchdog.cancel()
        if rank == 0:
            line["config"]["sharded_check"] = sharded_check
        emit()
    elif rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
