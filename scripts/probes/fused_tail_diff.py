import sys, os
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests", "golden"))
import numpy as np, torch, recipes as R
from summarizer_amd.models.vasnet import VASNet
torch.manual_seed(0)
lens = [int(np.ceil(v)) for v in np.random.default_rng(0).uniform(150, 320, 50)]
m = VASNet().cuda().eval()
x = torch.from_numpy(np.concatenate([R.features(T, 1, 1024, i)[:, 0, :] for i, T in enumerate(lens)])).cuda()
with torch.no_grad(): s = m.score_packed(x, lens).cpu().numpy()
np.save(sys.argv[1], s)
