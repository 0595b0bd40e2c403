import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)
def split(v):
    # v float32 array (already scaled); hi = truncate to 11 significant bits (mask 13 mantissa bits), lo = fp16(v-hi) (rtz approx by float16 cast)
    u = v.view(np.uint32) & np.uint32(0xffffe000)
    hi = u.view(np.float32)
    lo = (v - hi).astype(np.float16).astype(np.float32)
    return hi.astype(np.float16).astype(np.float32), lo
def pow2scale(mx):
    e = np.floor(np.log2(mx)); return np.float32(2.0 ** (12 - e))
BT = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], np.float32)
G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], np.float32)
AT = np.array([[1,1,1,0],[0,1,-1,-1]], np.float32)
def wino(x, w, acc64=False):
    # x [C,H,W] f32, w [O,C,3,3] f32 ; pad 1; returns [O,H,W]
    C,H,W = x.shape; O = w.shape[0]
    xp = np.zeros((C,H+2,W+2), np.float32); xp[:,1:-1,1:-1] = x
    U = np.einsum('ai,ocij,bj->aboc', G, w, G).astype(np.float32)       # [4,4,O,C]
    su = pow2scale(np.abs(U).max()); Uh, Ul = split(U*su)
    ty, tx = H//2, W//2
    d = np.zeros((ty,tx,C,4,4), np.float32)
    for i in range(4):
        for j in range(4):
            d[:,:,:,i,j] = xp[:, i:i+H:2, j:j+W:2].transpose(1,2,0)
    V = np.einsum('ai,yxcij,bj->abyxc', BT, d, BT).astype(np.float32)     # fp32 adds (einsum in f32)
    sv = pow2scale(np.abs(V).max()); Vh, Vl = split(V*sv)
    dt = np.float64 if acc64 else np.float32
    M = (np.einsum('aboc,abyxc->abyxo', Uh.astype(dt), Vh.astype(dt)) + np.einsum('aboc,abyxc->abyxo', Uh.astype(dt), Vl.astype(dt))
         + np.einsum('aboc,abyxc->abyxo', Ul.astype(dt), Vh.astype(dt))).astype(np.float32)
    Y = np.einsum('ia,abyxo,jb->yxoij', AT, M, AT).astype(np.float32) / (su*sv)
    out = np.zeros((O,H,W), np.float32)
    for i in range(2):
        for j in range(2):
            out[:, i::2, j::2] = Y[:,:,:,i,j].transpose(2,0,1)
    return out
def direct_split(x, w):
    C,H,W = x.shape; O=w.shape[0]
    sw = pow2scale(np.abs(w).max()); wh, wl = split(w*sw)
    sx = pow2scale(np.abs(x).max()); xh, xl = split(x*sx)
    f = lambda a,b: F.conv2d(torch.from_numpy(a)[None].double(), torch.from_numpy(b).double(), padding=1)[0]
    return ((f(xh,wh)+f(xh,wl)+f(xl,wh)).float().numpy())/(sw*sx)
for C,S,scale_x in [(64,32,1.0),(64,32,1e-3),(32,16,1.0)]:
    x = (torch.randn(C,S,S)*scale_x).numpy().astype(np.float32)
    w = (torch.randn(C,C,3,3)*3.0/(C*9)**.5).numpy().astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), padding=1)[0].numpy()
    f32 = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    for name, got in [('winograd split (f32 acc)', wino(x,w)), ('winograd split (f64 acc)', wino(x,w,True)), ('direct split', direct_split(x,w)), ('plain fp32 conv', f32)]:
        e = np.abs(got-ref)
        print(C,S,scale_x, name, 'max err / max|ref| = %.2e' % (e.max()/np.abs(ref).max()), ' rel-L2 = %.2e' % (np.linalg.norm(e)/np.linalg.norm(ref)))
