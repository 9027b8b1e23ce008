#!/usr/bin/env python
"""Headline benchmark: audio samples/s of `SynthesizerTrn.infer` on the
MI355X path, BASELINE.json configs[1] (ljs_mb_istft_vits, batch 64 per GPU,
T_text = 200, synthetic LJSpeech-length batch, synthetic checkpoint).

  python bench.py [--gpus N --steps K --warmup W]

`--gpus N` (N > 1) works as typed: when the process was not started by a launcher (no WORLD_SIZE in
the environment) it starts N ranks itself — `python -m torch.distributed.run --nproc-per-node N
bench.py …` as a CHILD process, before this process touches a GPU — relays rank 0's JSON line and
exits with the children's code.  Started under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.  On a box with fewer GPUs than ranks the
launcher falls back to a REHEARSAL (ranks share devices, `gloo` instead of RCCL) and the JSON says so.

A step = one full `infer` (text encoder .. waveform, all 8 reference outputs
materialised) over one batch of 64 utterances per GPU; with N > 1 the global
batch of 64 N utterances is sharded, every shard pads to the global T'max and
the waveforms are all-gathered (RCCL) inside the timed region.  `value` counts
VALID output samples (256 * sum y_lengths) of all ranks per second.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s float4 copy)
FP32_PEAK_TFLOPS = 157.3     # fp32 MFMA == fp32 vector peak
ISTFT_BYTES_PER_FRAME = 72 * 16 * 4 + 256 * 4           # 5632 B: SURVEY §8d (waveform-only mode)
ISTFT_BYTES_PER_FRAME_ALL = ISTFT_BYTES_PER_FRAME + 2 * 36 * 16 * 4 + 1024   # + spec, phase, o_mb (MB)
# BASELINE.json "configs": which entry a (config, per-GPU batch, n_gpus) combination is
BASELINE_CONFIGS = {("ljs_mini_mb_istft_vits", 1, 1): 0, ("ljs_mb_istft_vits", 64, 1): 1,
                    ("ljs_ms_istft_vits", 64, 1): 2, ("ljs_mb_istft_vits", 64, 8): 3,
                    ("uudb_ms_istft_vits_ms", 32, 8): 4}


def decoder_flops_per_frame(cfg):
    """2 * MAC of conv_pre .. subband_conv_post per z-frame (143.9 MFLOP for mb)."""
    C0, I = cfg.upsample_initial_channel, cfg.inter_channels
    mac = I * C0 * 7
    rate = 1
    for i in range(2):
        cin, cout = C0 >> i, C0 >> (i + 1)
        mac += cin * cout * 16 * rate            # ConvTranspose1d k16 stride u: 16/u taps x u outputs
        rate *= cfg.upsample_rates[i]
        mac += sum(6 * k for k in cfg.resblock_kernel_sizes) * cout * cout * rate
    mac += (C0 >> 2) * cfg.post_channels * 7 * rate
    return 2.0 * mac


def flow_flops_per_frame(cfg):
    """2 * MAC of the four reverse coupling layers per z-frame (14.2 MFLOP)."""
    H, I = cfg.hidden_channels, cfg.inter_channels
    per_flow = (I // 2) * H + 4 * (2 * H * H * 5) + 3 * (2 * H * H) + H * H + H * (I // 2)
    return 2.0 * 4 * per_flow


def encoder_flops_per_token(cfg, T):
    """2 * MAC of the text encoder per token at padded length T (13.5 MFLOP at T=200)."""
    H, Fc, I = cfg.hidden_channels, cfg.filter_channels, cfg.inter_channels
    per_layer = 4 * H * H + 2 * H * T + 2 * cfg.kernel_size * H * Fc
    return 2.0 * (cfg.n_layers * per_layer + 2 * I * H)


def kernel_source_sha(name):
    with open(os.path.join(ROOT, "mb-istft-vits_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def physical_cores():
    try:
        pairs = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or None
    except OSError:
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--t-text", type=int, default=200)
    ap.add_argument("--config", default="ljs_mb_istft_vits")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args(argv)


def self_launch(args):
    """N > 1 without a launcher: start the ranks as children of THIS process, which has not made a
    single GPU call (`torch.cuda.device_count()` does not initialise the GPU on this image; nothing
    else below touches it), relay their output and return their exit code."""
    import torch
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < 1:
        print("bench.py needs an MI355X (no CPU fallback exists for the product path)", file=sys.stderr)
        return 1
    if n_dev < args.gpus:
        # not a scaling measurement: the ranks share the devices that exist and talk over gloo
        print("bench.py: %d GPU(s) visible for --gpus %d -> REHEARSAL (ranks share devices, gloo backend); "
              "the JSON line is marked \"rehearsal\": true" % (n_dev, args.gpus), file=sys.stderr)
        env["MBV_BENCH_SHARE_DEVICES"] = str(n_dev)
        env.setdefault("MBV_BENCH_BACKEND", "gloo")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    share = int(os.environ.get("MBV_BENCH_SHARE_DEVICES", "0") or 0)
    if os.environ.get("MBV_BENCH_ONE_DEVICE"):       # older spelling of the rehearsal switch
        share = 1
    dev_index = local_rank % share if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    from mb_istft_vits_amd import models, utils, synth, dist as mdist, spec as mspec
    # MBV_BENCH_FORCE_DIST=1: take the sharded (RCCL) code path even with one rank — the rehearsal a
    # one-GPU box allows for broadcast / all-reduce / all-gather on the real backend
    dist_on = world > 1 or bool(os.environ.get("MBV_BENCH_FORCE_DIST"))
    backend = None
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("MBV_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    hps = utils.get_hparams_from_file(utils.builtin_config(args.config))
    net = models.SynthesizerTrn(59, hps.data.filter_length // 2 + 1,
                                hps.train.segment_size // hps.data.hop_length,
                                n_speakers=hps.data.n_speakers, **hps.model)
    cfg = net.cfg
    sr = hps.data.sampling_rate
    # ---- weights: rank 0 generates the checkpoint, folds and packs it once; the FOLDED arena is broadcast over
    # RCCL and imported by the other ranks (no state dict, no host-side fold, no upload there)
    if rank == 0:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 1234).items()})
    net = net.to(dev).eval()
    arena_floats, t_arena = None, None
    if dist_on:
        torch.cuda.synchronize(dev)
        t_a = time.perf_counter()
        arena_floats = mdist.broadcast_arena(net, src=0)
        torch.cuda.synchronize(dev)
        t_arena = time.perf_counter() - t_a

    B = args.batch
    x_np, xl_np, sid_np = synth.synthetic_batch(cfg, B * world, args.t_text, seed=0, ragged=args.ragged)
    x, xl = torch.from_numpy(x_np).to(dev), torch.from_numpy(xl_np).to(dev)
    sid = torch.from_numpy(sid_np).to(dev) if sid_np is not None else None

    # N > 1: the all-gathers of step k run on a side stream under the kernels of step k + 1 (`overlap="next"`: the
    # call returns a handle, its tensors are taken one step later); MBV_BENCH_OVERLAP=none|halves for A/B
    overlap = os.environ.get("MBV_BENCH_OVERLAP", "next")
    overlap = None if overlap in ("", "none") else overlap
    pending = []

    def step(outputs=None):
        # outputs=None: every shard / the single GPU materialises all 8 tensors of the reference tuple
        if dist_on:
            r = mdist.sharded_infer(net, x, xl, sid, noise_scale=0, length_scale=1, outputs=outputs, overlap=overlap)
            if isinstance(r, mdist.Gathered):
                pending.append(r)
                if len(pending) > 1:
                    return pending.pop(0).result()       # the step before this one: its gather has had a whole step
                return None, None
            o, ylen = r
        else:
            (o, *_), ylen = net.infer_with_lengths(x, xl, sid, noise_scale=0, length_scale=1, outputs=outputs)
        return o, ylen

    def drain():
        r = (None, None)
        while pending:
            r = pending.pop(0).result()
        return r

    def run_steps(n, outputs=None):
        """n steps back to back, every gather waited for at the end; -> the last step's (o, y_lengths)"""
        last = (None, None)
        for _ in range(n):
            r = step(outputs)
            if r[0] is not None:
                last = r
        r = drain()
        return r if r[0] is not None else last

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    o, ylen = run_steps(args.steps)              # exactly K steps, no host sync inside beyond infer's own; the last
    sync()                                       # step's gather is waited for inside the timed region
    elapsed = time.perf_counter() - t0
    # kernel-level timers for the roofline lines: a separate, untimed pass (reading the HIP events
    # synchronises the stream, which does not belong inside the timed region)
    conv_ms, istft_ms = [], []
    for _ in range(min(args.steps, 10)):
        run_steps(1)
        c, i = net.kernel_times_ms()             # HIP events on the launch stream
        conv_ms.append(c)
        istft_ms.append(i)
    sync()
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    valid_samples = int(ylen.sum().item()) * cfg.samples_per_frame      # whole job (ylen is global)
    Tp = o.shape[-1] // cfg.samples_per_frame
    value = valid_samples * args.steps / elapsed

    # secondary: the same job when the caller only takes the waveform (`outputs=("o",)`, what
    # tts_vits.py:134-137 uses of the tuple) — not the headline
    run_steps(2, ("o",))
    sync()
    t1 = time.perf_counter()
    n_wave = max(3, args.steps // 2)
    run_steps(n_wave, ("o",))
    sync()
    wave_elapsed = time.perf_counter() - t1
    if dist_on:
        t = torch.tensor([wave_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wave_elapsed = float(t.item())

    # ragged batches only: the OPT-IN trimmed decode (infer(outputs=("o",), trim=True): per utterance only what its valid
    # samples depend on is computed; those samples are bitwise the default's) — not the headline
    trim_info = None
    if args.ragged and not dist_on and cfg.decoder != mspec.DEC_SB:
        def tstep():
            return net.infer(x, xl, sid, noise_scale=0, length_scale=1, outputs=("o",), trim=True)[0]
        for _ in range(2):
            o_t = tstep()
        sync()
        t3 = time.perf_counter()
        for _ in range(n_wave):
            o_t = tstep()
        sync()
        trim_elapsed = time.perf_counter() - t3
        n_valid = [int(v) * cfg.samples_per_frame for v in ylen.tolist()]
        same = all(bool(torch.equal(o_t[b, 0, :n], o[b, 0, :n])) for b, n in enumerate(n_valid))
        trim_info = {"what": "OPT-IN infer(outputs=('o',), trim=True) on this ragged batch: decoder / iSTFT tiles behind "
                             "y_lengths[b] + 32 frames are not computed; valid samples bitwise the default's",
                     "value": round(valid_samples * n_wave / trim_elapsed, 1), "ms_per_step": round(trim_elapsed / n_wave * 1e3, 3),
                     "valid_samples_bitwise_equal_to_default": same,
                     "padded_frame_share": round(1.0 - float(ylen.sum()) / (len(n_valid) * (o.shape[-1] // cfg.samples_per_frame)), 4)}

    # tertiary: the OPT-IN split-bf16 conv mode (conv_bf16 = 3; not IEEE fp32 multiplication, so never the
    # headline): same job, all outputs, plus its waveform's distance from the exact mode's
    o_exact = o.clone()
    net.set_option("conv_bf16", 3)
    try:
        run_steps(2)
        sync()
        t2 = time.perf_counter()
        o3, _ = run_steps(n_wave)
        sync()
        bf16_elapsed = time.perf_counter() - t2
        bf16_rms = float((o3.double() - o_exact.double()).pow(2).mean().sqrt())
    finally:
        net.set_option("conv_bf16", 0)
    run_steps(2)                                 # back on the exact path before the per-stage timers below are read
    sync()
    if dist_on:
        t = torch.tensor([bf16_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bf16_elapsed = float(t.item())

    dist_info = None
    if dist_on:
        names = [None] * world
        dist.all_gather_object(names, "rank %d: cuda:%d %s" % (rank, dev_index, torch.cuda.get_device_name(dev)))
        # where a step's time goes, from HIP events on the streams the work was issued on (untimed pass; MAX over
        # ranks): the plain schedule (gathers on the caller's stream) gives each collective's own time, the
        # overlapped one what is left exposed
        def timed(ov, n=4):
            acc = None
            for _ in range(n):
                tm = mdist.StepTimes()
                r = mdist.sharded_infer(net, x, xl, sid, noise_scale=0, length_scale=1, outputs=None, overlap=ov, timing=tm)
                if isinstance(r, mdist.Gathered):
                    r.result()
                torch.cuda.synchronize(dev)
                d = tm.ms()
                acc = d if acc is None else {k: (acc[k] + v if isinstance(v, float) else v) for k, v in d.items()}
            vals = torch.tensor([acc[k] / n for k in sorted(acc) if isinstance(acc[k], float)], device=dev, dtype=torch.float64)
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            return {k: round(float(v), 4) for k, v in zip([k for k in sorted(acc) if isinstance(acc[k], float)], vals.tolist())}
        timing_ms = {"plain": timed(None)}
        if overlap:
            timing_ms[overlap] = timed(overlap)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": names,
                     "rehearsal": bool(share), "overlap": overlap, "collectives_per_step":
                     "1 all_reduce(MAX) of [T', status] (2 x int64) + all_gather(waveform rows) + all_gather(y_lengths)",
                     "timing_ms": timing_ms,
                     "weights": {"what": "rank 0 folds once, broadcast of the folded arena, mbv_import_arena on the others",
                                 "arena_mb": round(arena_floats * 4 / 1e6, 1), "broadcast_and_import_s": round(t_arena, 3)}}

    stage_ms = None
    if rank == 0:
        out = net.infer(x[:B], xl[:B], sid[:B] if sid is not None else None, noise_scale=0, length_scale=1)
        stage_ms = {k: round(v * 1e3, 3) for k, v in dict(out[7]).items()}

    # ---- roofline of the fused iSTFT+PQMF launch ------------------------
    roof, roof_conv, roof_other = None, None, None
    if rank == 0 and cfg.decoder != mspec.DEC_SB:
        from mb_istft_vits_amd.benchutil import istft_waveform_only_ms
        frames = B * Tp
        wave_ms = istft_waveform_only_ms(net, B, Tp, iters=50)
        # the same launch on a working set far past the 256 MiB Infinity Cache: rotate enough
        # distinct (input, output) buffer sets that nothing is still cached when a set comes round again
        set_bytes = ISTFT_BYTES_PER_FRAME * frames
        nsets = max(3, int(np.ceil(3.0 * 256 * 2 ** 20 / set_bytes)))
        cold_ms = istft_waveform_only_ms(net, B, Tp, iters=50, rotate=nsets)
        all_ms = float(np.median(istft_ms))
        ach = set_bytes / (wave_ms * 1e-3) / 1e9
        ach_cold = set_bytes / (cold_ms * 1e-3) / 1e9
        ach_all = ISTFT_BYTES_PER_FRAME_ALL * frames / (all_ms * 1e-3) / 1e9
        # HBM traffic from PMC counters: a separate rocprofv3 --pmc pass (it cannot run inside this
        # process); only quoted when it was taken for THIS kernel source and THIS launch shape
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "istft_pqmf_pmc.json")
        if os.path.isfile(pmc):
            pj = json.load(open(pmc))
            if (pj.get("B"), pj.get("Tp")) == (B, Tp) and pj.get("kernel_source_sha16") == kernel_source_sha("istft_pqmf.hip"):
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_src = "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel source at this shape (%s); not this run" % pj.get("profile", "profiles/")
        roof = {"kernel": "istft_pqmf_kernel<480,512> (waveform-only)", "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": set_bytes, "ms_per_launch": round(wave_ms, 4),
                "note": "working set %.0f MB < 256 MiB Infinity Cache: as in the pipeline, where x_post was just written by the preceding conv" % (set_bytes / 1e6),
                "past_cache": {"what": "same launch, %d rotating buffer sets (%.0f MB working set)" % (nsets, nsets * set_bytes / 1e6),
                               "ms_per_launch": round(cold_ms, 4), "achieved": round(ach_cold, 1),
                               "frac": round(ach_cold / HBM_PEAK_GBS, 4), "unit": "GB/s"},
                "all_outputs": {"what": "the launch `infer` issues by default (spec, phase, o_mb written too: %d B/frame)" % ISTFT_BYTES_PER_FRAME_ALL,
                                "ms_per_launch": round(all_ms, 4), "achieved": round(ach_all, 1),
                                "frac": round(ach_all / HBM_PEAK_GBS, 4), "unit": "GB/s"}}
    if rank == 0:
        frames = B * Tp
        fl = decoder_flops_per_frame(cfg) * frames
        # median of the per-step HIP-event readings (r03z: one stalled step of ten moved the MEAN by 2 ms, 0.82 -> 0.78,
        # while the timed region next to it and the run after it on the same box read 40.5 ms); min / max printed beside it
        cm = float(np.median(conv_ms))
        roof_conv = {"kernel": "decoder conv stack (conv1d_mfma, fp32 MFMA)", "bound": "mfma",
                     "achieved": round(fl / (cm * 1e-3) / 1e12, 2), "peak": FP32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(fl / (cm * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "ms_per_step": round(cm, 3), "ms_per_step_min_max": [round(float(np.min(conv_ms)), 3), round(float(np.max(conv_ms)), 3)],
                     "samples": len(conv_ms), "flop_per_step": fl}
        pmc_conv = os.path.join(ROOT, "profiles", "conv_mfma_pmc.json")
        if os.path.isfile(pmc_conv):             # matrix-pipe busy share from a separate --pmc pass
            pj = json.load(open(pmc_conv))
            if pj.get("kernel_source_sha16") == kernel_source_sha("conv1d.hip"):
                roof_conv["mfma_util_percent_pmc"] = pj.get("mfma_util_percent")
                roof_conv["mfma_util_source"] = pj.get("source")
        if stage_ms:
            fe = encoder_flops_per_token(cfg, args.t_text) * B * args.t_text
            ff = flow_flops_per_frame(cfg) * frames
            valid_frac = valid_samples / float(cfg.samples_per_frame * frames * world)
            roof_other = {
                "what": "stage FLOPs over PADDED positions (SURVEY 8d: 13.5 MFLOP/token, 14.2 MFLOP/frame) / HIP-event stage time / 157.3 TFLOP/s",
                "text_encoder": {"tflops": round(fe / (stage_ms["text_encoder"] * 1e-3) / 1e12, 1),
                                 "frac": round(fe / (stage_ms["text_encoder"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 3)},
                "flow": {"tflops": round(ff / (stage_ms["flow"] * 1e-3) / 1e12, 1),
                         "frac": round(ff / (stage_ms["flow"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 3),
                         "note": "the fused WN kernel does not compute masked frames (%.1f %% of the padded frames are valid): "
                                 "executed FLOP/s = %.1f TFLOP/s" % (100 * valid_frac, valid_frac * ff / (stage_ms["flow"] * 1e-3) / 1e12)}}

    # ---- CPU baseline: the oracle ("port") on this box's host cores ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_infer
        nb = min(8, B)
        sd_np = synth.make_state_dict(cfg, 1234)
        W = ref_infer.Weights(sd_np)
        sid8 = sid_np[:nb] if sid_np is not None else None

        def cpu_run(threads):
            torch.set_num_threads(threads)
            ref_infer.infer(W, cfg, x_np[:2], xl_np[:2], sid8[:2] if sid8 is not None else None)   # warm-up
            ts = []
            for _ in range(3):
                t1 = time.perf_counter()
                r = ref_infer.infer(W, cfg, x_np[:nb], xl_np[:nb], sid8)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            return ts[1], int(r["y_lengths"].sum()) * cfg.samples_per_frame        # median of 3

        # the port does not scale past a few dozen threads at this problem size (oneDNN conv over
        # 8 utterances): time a few thread counts and report the FASTEST as the baseline
        n_all = max(1, min(os.cpu_count() or 1, 64))
        by_threads = {}
        cpu_samples = 0
        for n in sorted({min(8, n_all), min(16, n_all), min(32, n_all), n_all}):
            t_n, cpu_samples = cpu_run(n)
            by_threads[n] = t_n
        best = min(by_threads, key=by_threads.get)
        t_best = by_threads[best]
        cpu_model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": round(cpu_samples / t_best, 1), "unit": "samples/s", "cores": best, "kind": "port",
               "cores_meaning": "threads used (torch.set_num_threads) at the fastest of the thread counts tried",
               "physical_cores": physical_cores(), "logical_cpus": os.cpu_count(),
               "sample": "first %d utterances of the batch-%d workload; oracle (PyTorch-CPU fp32 restatement "
                         "of the reference) infer, 1 warm-up (B=2) + median of 3 timed calls per thread "
                         "count; value = the fastest thread count" % (nb, B),
               "rtf": round(t_best / (cpu_samples / sr), 5),
               "samples_per_s_by_threads": {str(n): round(cpu_samples / t, 1) for n, t in by_threads.items()},
               "cpu_model": cpu_model, "torch": torch.__version__}

    if rank == 0:
        idx = BASELINE_CONFIGS.get((args.config, B, world))
        label = ("configs[%d]" % idx) if idx is not None and args.t_text == 200 and not args.ragged else \
            "not a BASELINE.json config"
        line = {
            "metric": "audio samples/sec (%g kHz), MB-iSTFT-VITS infer, batch %d per GPU" % (sr / 1000.0, B),
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (token ids uniform, T_text=%d%s; synthetic checkpoint seed 1234)" %
                    (args.t_text, " ragged" if args.ragged else ""),
            "config": {"workload": "%s: %s, batch %d per GPU, infer(noise_scale=0, length_scale=1), all 8 outputs"
                                   % (label, args.config, B),
                       "global_batch": B * world, "t_text": args.t_text, "t_frames_max": Tp,
                       "valid_samples_per_step": valid_samples, "sampling_rate": sr,
                       "parallelism": "utterance-sharded dp%d" % world},
            "rtf": round((elapsed / args.steps) / (valid_samples / sr), 7),
            "waveform_only": {"what": "same job with infer(outputs=('o',)) (a caller that takes [0] only)",
                              "value": round(valid_samples * n_wave / wave_elapsed, 1),
                              "ms_per_step": round(wave_elapsed / n_wave * 1e3, 3)},
            "conv_bf16": {"what": "OPT-IN mode mbv_set_option('conv_bf16', 3): split-bf16 operands (hi + mid planes, three products, "
                                  "fp32 accumulation) in the large conv launches; all 8 outputs; NOT the headline (the headline is exact fp32)",
                          "value": round(valid_samples * n_wave / bf16_elapsed, 1),
                          "ms_per_step": round(bf16_elapsed / n_wave * 1e3, 3),
                          "waveform_rms_vs_exact_mode": bf16_rms, "bar": 1e-4},
            "trim": trim_info,
            "dist": dist_info,
            "stage_ms": stage_ms, "roofline": roof, "roofline_conv": roof_conv, "roofline_other": roof_other,
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu_rtf"] = round(cpu["rtf"] / line["rtf"], 1)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
