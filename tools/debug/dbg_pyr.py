import numpy as np, sys, os
sys.path.insert(0, '/root/repo')
from dani_slam_amd import synth, frontend
from oracle import oracle
w = synth.superpoint_weights(0)
img = synth.frame(1)
e = frontend.Engine(device=0, max_batch=1, weights=w)
r = e.sp_extract_batch(img[None], nfeatures=1000, nlevels=4)[0]
ref = oracle.sp_extract(img, w, nfeatures=1000, nlevels=4)
per, sc = oracle.features_per_level(1000,1.2,4)
cur = img; off = 0
for l in range(4):
    if l:
        inv = np.float32(1.0)/sc[l]
        nw, nh = int(np.rint(np.float32(640)*inv)), int(np.rint(np.float32(480)*inv))
        cur = oracle.resize_bilinear_u8c1(cur, nw, nh)
    h, wd = cur.shape
    prob_o, desc_o = oracle.superpoint_forward(cur, w)
    cand = oracle.collect_candidates(prob_o, wd, h, 0.015, 0.007, (), float(sc[l]))
    keys = oracle.distribute_octtree(cand, 16, wd-16, 16, h-16, int(per[l]))
    kxy = keys[:, :2] + 16
    d = oracle.compute_descriptors(desc_o, kxy, (h//8)*8, (wd//8)*8)
    n = len(kxy)
    g = r["desc"][off:off+n]; o = ref["desc"][off:off+n]
    print("level", l, n, "py-oracle vs C-oracle eq", np.array_equal(d, o), "gpu vs C-oracle eq", np.array_equal(g, o), "gpu vs py eq", np.array_equal(g, d))
    if not np.array_equal(g, o):
        dd = np.abs(g-o); cols = np.nonzero(dd.max(0)>0)[0]
        # column norms
        print("   gpu col norm", np.linalg.norm(g.astype(np.float64),axis=0)[:4], "oracle", np.linalg.norm(o.astype(np.float64),axis=0)[:4])
        ratio = g[:, cols[0]]/o[:, cols[0]]; print("   ratio col", cols[0], ratio[:5], "int coords?", np.all(kxy==np.rint(kxy)), kxy[:3])
    off += n
os.environ["DANI_DEBUG_LEVEL"] = "3"
prob, dmap, candg = e.sp_debug_fetch(0, 278, 370)
print("cand eq", candg.shape, cand.shape, np.array_equal(candg, cand))
if candg.shape == cand.shape:
    dif = np.nonzero((candg != cand).any(1))[0]
    print("n differing cand", len(dif)); print(candg[dif[:5]], cand[dif[:5]])
    nonint_g = (candg[:, :2] != np.rint(candg[:, :2])).any(1).sum(); nonint_o = (cand[:, :2] != np.rint(cand[:, :2])).any(1).sum()
    print("non-integer coords gpu", nonint_g, "oracle", nonint_o)
