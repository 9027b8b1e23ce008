import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpu_util import make_net
from mb_istft_vits_amd import synth
net, sd = make_net("ljs_mini_mb_istft_vits")
for B, T, seed in ((1, 30, 1), (2, 30, 2), (2, 31, 3), (3, 20, 4), (4, 25, 5)):
    x, xl, _ = synth.synthetic_batch(net.cfg, B, T, seed=seed, ragged=True)
    xg, xlg = torch.from_numpy(x).cuda(), torch.from_numpy(xl).cuda()
    net.set_option("wn_fused", 0)
    (o0, *_r0), yl = net.infer_with_lengths(xg, xlg, noise_scale=0, length_scale=1)
    z0 = _r0[5][0]
    net.set_option("wn_fused", 1)
    (o1, *_r1), yl = net.infer_with_lengths(xg, xlg, noise_scale=0, length_scale=1)
    z1 = _r1[5][0]
    yl = yl.tolist()
    hs = np.cumsum([0] + [(v + 15) // 16 for v in yl])
    print("B=%d ylen %s halfunit starts %s" % (B, yl, hs.tolist()))
    d = (z1 - z0).abs()
    for b in range(B):
        db = d[b].max(0).values.cpu().numpy()
        bad = np.nonzero(db > 1e-4)[0]
        print("   utt %d max diff %.2e first bad frame %s n_bad %d" % (b, db.max(), bad[:1], len(bad)))
