"""debug: stage-by-stage agreement of the bf16 arithmetic (plane kernels, inference, one video) with bf16-rounded-operand matmuls"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import recipes as R
from summarizer_amd import kernels
from summarizer_amd.models.vasnet import VASNet
dev = torch.device("cuda:0")
D, T = 1024, 300
w = R.vasnet_weights(D, 41)
m = VASNet(input_size=D, precision="bf16"); m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev).eval()
x = torch.from_numpy((R.features(T, 1, D, 5000) - 0.1)[:, 0, :]).to(dev)
sb = kernels.SeqBatch.get([T], dev)
r16 = lambda t: t.to(torch.bfloat16).to(torch.float64)
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
    Q = r16(x) @ r16(p["Q.weight"]).t(); K = r16(x) @ r16(p["K.weight"]).t(); V = r16(x) @ r16(p["V.weight"]).t()
    ref = torch.cat([Q, K, V], 1)
    print("QKV vs bf16-operand product:", float((qkv.double() - ref).abs().max()), " vs fp64 product:", float((qkv.double() - torch.cat([x.double() @ p[k].double().t() for k in ("Q.weight", "K.weight", "V.weight")], 1)).abs().max()), "max", float(ref.abs().max()))
    e = r16(qkv[:, :D]) @ r16(qkv[:, D:2 * D]).t() * float(m.scale)
    a = torch.softmax(e, 1)
    print("alpha vs emu:", float((E[:, :T].double() - a).abs().max()), " vs fp32-operand:", float((E[:, :T].double() - torch.softmax(qkv[:, :D].double() @ qkv[:, D:2 * D].double().t() * float(m.scale), 1)).abs().max()))
    c = r16(E[:, :T]) @ r16(qkv[:, 2 * D:])
    print("ctx vs emu:", float((ctx.double() - c).abs().max()), " vs fp32-operand:", float((ctx.double() - E[:, :T].double() @ qkv[:, 2 * D:].double()).abs().max()), "max", float(c.abs().max()))
    y = r16(ctx) @ r16(p["attention_head_projection.weight"]).t() + x.double()
    print("y0 vs emu:", float((y0.double() - y).abs().max()), " vs fp32-operand:", float((y0.double() - (ctx.double() @ p["attention_head_projection.weight"].double().t() + x.double())).abs().max()))
    zz = torch.relu(r16(y1) @ r16(p["k1.weight"]).t() + p["k1.bias"].double())
    print("z vs emu:", float((z.double() - zz).abs().max()), " vs fp32-operand:", float((z.double() - torch.relu(y1.double() @ p["k1.weight"].double().t() + p["k1.bias"].double())).abs().max()))
