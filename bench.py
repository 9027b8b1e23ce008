#!/usr/bin/env python
"""Headline benchmark: audio samples/s of `SynthesizerTrn.infer` on the
MI355X path, BASELINE.json configs[1] (ljs_mb_istft_vits, batch 64 per GPU,
T_text = 200, synthetic LJSpeech-length batch, synthetic checkpoint).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one full `infer` (text encoder .. waveform, all 8 reference outputs
materialised) over one batch of 64 utterances per GPU; with N > 1 the global
batch of 64 N utterances is sharded, every shard pads to the global T'max and
the waveforms are all-gathered (RCCL) inside the timed region.  `value` counts
VALID output samples (256 * sum y_lengths) of all ranks per second.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s float4 copy)
FP32_PEAK_TFLOPS = 157.3     # fp32 MFMA == fp32 vector peak
ISTFT_BYTES_PER_FRAME = 72 * 16 * 4 + 256 * 4           # 5632 B: SURVEY §8d (waveform-only mode)
ISTFT_BYTES_PER_FRAME_ALL = ISTFT_BYTES_PER_FRAME + 2 * 36 * 16 * 4 + 1024   # + spec, phase, o_mb (MB)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--t-text", type=int, default=200)
    ap.add_argument("--config", default="ljs_mb_istft_vits")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" %
                             (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    if os.environ.get("MBV_BENCH_ONE_DEVICE"):       # rehearsal: all ranks on cuda:0 (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from mb_istft_vits_amd import models, utils, synth, dist as mdist, spec as mspec
    # MBV_BENCH_FORCE_DIST=1: take the sharded (RCCL) code path even with one rank — the rehearsal a
    # one-GPU box allows for broadcast / all-reduce / all-gather on the real backend
    dist_on = world > 1 or bool(os.environ.get("MBV_BENCH_FORCE_DIST"))
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
    # ---- weights: rank 0 generates, RCCL broadcast to the rest -----------------
    sd_np = synth.make_state_dict(cfg, 1234) if rank == 0 else None
    if dist_on:
        shapes = mspec.param_shapes(cfg)
        sd = mdist.broadcast_state_dict(
            {k: torch.from_numpy(v) for k, v in sd_np.items()} if rank == 0 else None, shapes, dev)
        net.load_state_dict(sd)
    else:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    net = net.to(dev).eval()

    B = args.batch
    x_np, xl_np, sid_np = synth.synthetic_batch(cfg, B * world, args.t_text, seed=0, ragged=args.ragged)
    x, xl = torch.from_numpy(x_np).to(dev), torch.from_numpy(xl_np).to(dev)
    sid = torch.from_numpy(sid_np).to(dev) if sid_np is not None else None

    def step():
        if dist_on:
            o, ylen = mdist.sharded_infer(net, x, xl, sid, noise_scale=0, length_scale=1)
        else:
            (o, *_), ylen = net.infer_with_lengths(x, xl, sid, noise_scale=0, length_scale=1)
        return o, ylen

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        o, ylen = step()
    sync()
    conv_ms, istft_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o, ylen = step()
        c, i = net.kernel_times_ms()             # HIP events on the launch stream (syncs this step)
        conv_ms.append(c)
        istft_ms.append(i)
    sync()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    valid_samples = int(ylen.sum().item()) * cfg.samples_per_frame      # whole job (ylen is global)
    Tp = o.shape[-1] // cfg.samples_per_frame
    value = valid_samples * args.steps / elapsed

    stage_ms = None
    if rank == 0:
        out = net.infer(x[:B], xl[:B], sid[:B] if sid is not None else None, noise_scale=0, length_scale=1)
        stage_ms = {k: round(v * 1e3, 3) for k, v in dict(out[7]).items()}

    # ---- roofline of the fused iSTFT+PQMF launch, waveform-only mode ----
    roof, roof_conv = None, None
    if rank == 0:
        from mb_istft_vits_amd.benchutil import istft_waveform_only_ms
        wave_ms = istft_waveform_only_ms(net, B, Tp, iters=50)
        frames = B * Tp
        ach = ISTFT_BYTES_PER_FRAME * frames / (wave_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "istft_pqmf_pmc.json")
        if os.path.isfile(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        roof = {"kernel": "istft_pqmf_kernel<480,512> (waveform-only)", "bound": "hbm",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": ISTFT_BYTES_PER_FRAME * frames, "ms_per_launch": round(wave_ms, 4),
                "all_outputs": {"ms_per_launch": round(float(np.mean(istft_ms)), 4),
                                "achieved": round(ISTFT_BYTES_PER_FRAME_ALL * frames /
                                                  (float(np.mean(istft_ms)) * 1e-3) / 1e9, 1),
                                "unit": "GB/s"}}
        fl = decoder_flops_per_frame(cfg) * frames
        cm = float(np.mean(conv_ms))
        roof_conv = {"kernel": "decoder conv stack (conv1d_mfma/convt4_mfma, fp32 MFMA)", "bound": "mfma",
                     "achieved": round(fl / (cm * 1e-3) / 1e12, 2), "peak": FP32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(fl / (cm * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "ms_per_step": round(cm, 3), "flop_per_step": fl}
        pmc_conv = os.path.join(ROOT, "profiles", "conv_mfma_pmc.json")
        if os.path.isfile(pmc_conv):             # matrix-pipe busy share from a separate --pmc pass
            roof_conv["mfma_util_percent_pmc"] = json.load(open(pmc_conv)).get("mfma_util_percent")

    # ---- CPU baseline: the oracle ("port") on this box's host cores ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_infer
        nb = 8
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
               "sample": "first %d utterances of the batch-%d workload; oracle (PyTorch-CPU fp32 restatement "
                         "of the reference) infer, 1 warm-up (B=2) + median of 3 timed calls per thread "
                         "count; value = the fastest thread count" % (nb, B),
               "rtf": round(t_best / (cpu_samples / sr), 5),
               "samples_per_s_by_threads": {str(n): round(cpu_samples / t, 1) for n, t in by_threads.items()},
               "cpu_model": cpu_model, "torch": torch.__version__}

    if rank == 0:
        line = {
            "metric": "audio samples/sec (%g kHz), MB-iSTFT-VITS infer, batch %d per GPU" % (sr / 1000.0, B),
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (token ids uniform, T_text=%d%s; synthetic checkpoint seed 1234)" %
                    (args.t_text, " ragged" if args.ragged else ""),
            "config": {"workload": "configs[1]: %s, batch %d per GPU, infer(noise_scale=0, length_scale=1)"
                                   % (args.config, B),
                       "global_batch": B * world, "t_text": args.t_text, "t_frames_max": Tp,
                       "valid_samples_per_step": valid_samples, "sampling_rate": sr,
                       "parallelism": "utterance-sharded dp%d" % world},
            "rtf": round((elapsed / args.steps) / (valid_samples / sr), 7),
            "stage_ms": stage_ms, "roofline": roof, "roofline_conv": roof_conv, "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu_rtf"] = round(cpu["rtf"] / line["rtf"], 1)
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
