"""debug: intermediates of the small-batch forward against torch matmuls on the GPU"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import kernels
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0")
D, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 300
w = R.vasnet_weights(D, 7000)
m = VASNet(input_size=D); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
x = torch.from_numpy(R.features(T, 1, D, 7100)[:, 0, :]).to(dev)
sb = kernels.SeqBatch.get([T], dev)
with torch.no_grad():
    s, _ = kernels.vasnet_forward_packed(x, sb, m._params(), m._opts(False), None, None, training=False)
    torch.cuda.synchronize()
    ws = kernels._ws_cache[(str(dev), torch.cuda.current_stream(dev).cuda_stream)]
    al = lambda v: (v + 255) // 256 * 256
    o = 0
    qkv = ws[o:o + T * 3 * D * 4].view(torch.float32).view(T, 3 * D); o += al(T * 3 * D * 4)
    ldE = (T + 3) // 4 * 4
    E = ws[o:o + T * ldE * 4].view(torch.float32).view(T, ldE); o += al(T * ldE * 4)
    ctx = ws[o:o + T * D * 4].view(torch.float32).view(T, D); o += al(T * D * 4)
    y0 = ws[o:o + T * D * 4].view(torch.float32).view(T, D); o += al(T * D * 4)
    y1 = ws[o:o + T * D * 4].view(torch.float32).view(T, D); o += al(T * D * 4)
    z = ws[o:o + T * D * 4].view(torch.float32).view(T, D); o += al(T * D * 4)
    p = m._params()
    xd = x.double()
    Q = xd @ p["Q.weight"].double().t(); K = xd @ p["K.weight"].double().t(); V = xd @ p["V.weight"].double().t()
    ref_qkv = torch.cat([Q, K, V], 1)
    print("QKV err", float((qkv.double() - ref_qkv).abs().max()), "max", float(ref_qkv.abs().max()))
    for g, nm in enumerate("QKV"):
        d = (qkv[:, g * D:(g + 1) * D].double() - ref_qkv[:, g * D:(g + 1) * D]).abs()
        print(" ", nm, float(d.max()), "bad rows", int((d.max(1).values > 1e-3).sum()), "bad cols", int((d.max(0).values > 1e-3).sum()))
    a = torch.softmax(qkv[:, :D].double() @ qkv[:, D:2 * D].double().t() * float(m.scale), dim=1)
    print("alpha err", float((E[:, :T].double() - a).abs().max()))
    c = E[:, :T].double() @ qkv[:, 2 * D:].double()
    print("ctx err", float((ctx.double() - c).abs().max()), float(c.abs().max()))
    y = ctx.double() @ p["attention_head_projection.weight"].double().t() + xd
    print("y0 err", float((y0.double() - y).abs().max()), float(y.abs().max()))
    zz = torch.relu(y1.double() @ p["k1.weight"].double().t() + p["k1.bias"].double())
    print("z err", float((z.double() - zz).abs().max()), float(zz.abs().max()))
    print("scores", s[:5].tolist())
    d = (qkv.double() - ref_qkv).abs()
    bad = (d > 1e-3)
    print("bad by row (first 70):", "".join("x" if b else "." for b in bad[:70, :].any(1).tolist()))
    print("bad by col (first 140):", "".join("x" if b else "." for b in bad[:, :140].any(0).tolist()))
    print("row 0 bad cols count", int(bad[0].sum()), "row 5 bad cols", int(bad[5].sum()), "of", 3 * D)
    r5 = qkv[5, :8].tolist(); print("row5 got", r5, "want", ref_qkv[5, :8].tolist())
    # is a wrong row equal to some partial sum?  compare with K-slice partial sums
    for nsl in (2, 4, 8):
        kc = D // nsl
        part = [xd[:, i * kc:(i + 1) * kc] @ p["Q.weight"].double()[:, i * kc:(i + 1) * kc].t() for i in range(nsl)]
        for i in range(nsl):
            e = float((qkv[5, :64].double() - part[i][5, :64]).abs().max())
            if e < 1e-3: print("row 5 equals partial", i, "of", nsl)
        cum = part[0].clone()
        for i in range(1, nsl):
            cum += part[i]
            e = float((qkv[5, :64].double() - cum[5, :64]).abs().max())
            if e < 1e-3: print("row 5 equals sum of first", i + 1, "of", nsl)
