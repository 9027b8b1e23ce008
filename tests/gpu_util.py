"""Helpers for the -m gpu parity tests: build models on the GPU from the synthetic checkpoint."""
import ctypes as C

import numpy as np
import torch

from mb_istft_vits_amd import models, synth, utils as mutils, _capi


def make_net(cfg_name, n_vocab=59, seed=1234, device="cuda:0", overrides=None):
    hps = mutils.get_hparams_from_file(mutils.builtin_config(cfg_name))
    for k, v in (overrides or {}).items():
        hps.model[k] = v
    net = models.SynthesizerTrn(n_vocab, hps.data.filter_length // 2 + 1,
                                hps.train.segment_size // hps.data.hop_length,
                                n_speakers=hps.data.n_speakers, **hps.model)
    sd = synth.make_state_dict(net.cfg, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net.to(device).eval(), sd


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def op_conv1d(net, x, w, bias, K, dil, slope):
    """x cuda [B,Cin,T]; w/bias numpy"""
    h = net._ensure_handle()
    B, Cin, T = x.shape
    Cout = w.shape[0]
    y = torch.empty(B, Cout, T, device=x.device, dtype=torch.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
    rc = _capi.lib().mbv_op_conv1d(h, ptr(x), w.ctypes.data_as(C.c_void_p),
                                   b.ctypes.data_as(C.c_void_p) if b is not None else None, ptr(y),
                                   B, Cin, Cout, T, K, dil, float(slope), net._stream())
    _capi.check(h, rc, "mbv_op_conv1d")
    return y


def op_istft_pqmf(net, x_post, filt=None, multistream=False, extras=True):
    h = net._ensure_handle()
    B, _, Fr = x_post.shape
    Tp = (Fr - 1) // 16
    dev = x_post.device
    o = torch.empty(B, 1, 256 * Tp, device=dev)
    o_mb = torch.empty(B, 4, (256 if multistream else 64) * Tp, device=dev) if extras else None
    spec = torch.empty(B, 4, 9, Fr, device=dev) if extras else None
    phase = torch.empty(B, 4, 9, Fr, device=dev) if extras else None
    rc = _capi.lib().mbv_istft_pqmf(h, ptr(x_post), B, Tp, ptr(filt), int(multistream), ptr(o),
                                    ptr(o_mb), ptr(spec), ptr(phase), net._stream())
    _capi.check(h, rc, "mbv_istft_pqmf")
    return o, o_mb, spec, phase
