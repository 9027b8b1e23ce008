import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import make_net
from mb_istft_vits_amd.benchutil import istft_waveform_only_ms
net = make_net("ljs_mini_mb_istft_vits")[0]
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("MBV_"))
for rot in (1, 4):
    ms = istft_waveform_only_ms(net, 64, 566, iters=50, rotate=rot)
    print("[%s] waveform-only rotate=%d: %.2f us -> %.0f GB/s frac %.3f" % (tag, rot, ms * 1e3, 5632 * 64 * 566 / (ms * 1e-3) / 1e9, 5632 * 64 * 566 / (ms * 1e-3) / 8e12))
