import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
from mb_istft_vits_amd.benchutil import istft_waveform_only_ms
net, sd = make_net("ljs_mb_istft_vits")
B, Tp = 64, 566
print("fresh w10  : %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50, warm=10))
time.sleep(2.0)
print("idle w1000 : %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50))
time.sleep(2.0)
print("idle w300  : %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50, warm=300))
x, xl, _ = synth.synthetic_batch(net.cfg, 64, 200, seed=0)
xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
for _ in range(5):
    o = net.infer(xg, xlg, noise_scale=0, length_scale=1)[0]
torch.cuda.synchronize()
print("after infer: %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50))
print("iters 200  : %.4f" % istft_waveform_only_ms(net, B, Tp, iters=200))
time.sleep(2.0)
print("after sleep: %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50))
del o
torch.cuda.empty_cache()
print("empty cache: %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50))
for _ in range(3):
    o = net.infer(xg, xlg, noise_scale=0, length_scale=1)[0]
torch.cuda.synchronize()
print("after infer: %.4f" % istft_waveform_only_ms(net, B, Tp, iters=50))
