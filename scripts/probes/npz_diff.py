import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for k in a.files:
    d=np.abs(a[k]-b[k]).max(); m=np.abs(b[k]).max()
    rel=np.linalg.norm((a[k]-b[k]).ravel())/max(np.linalg.norm(b[k].ravel()),1e-30)
    print(f"{k:45s} max|d| {d:.3e} max|b| {m:.3e} relL2 {rel:.3e} finite {np.isfinite(a[k]).all()}")
