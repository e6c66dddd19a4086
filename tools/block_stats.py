"""TEST-SIDE TOOL (uses the CPU oracle): how many shading passes (distinct draws) and distinct triangles an 8x8 block of the BASELINE scenes holds -- what a per-triangle
attribute setup in the resolve could be amortised over (verdict item 5).  usage: python tools/block_stats.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge
m = ge.load_package()
import oracle_binding as ob
for name, mk in (("c5", m.scenes.box_hall), ("c3", m.scenes.displaced_sphere), ("c4", m.scenes.heightfield_grid)):
    scene = mk()
    r = ob.render(scene, nthreads=8, want_bgra8=False)
    prim = r["prim"]
    bases = np.cumsum([0] + [d.num_triangles for d in scene.draws])
    draw = np.searchsorted(bases, prim, side="right") - 1
    draw[prim == 0xFFFFFFFF] = -1
    H, W = prim.shape
    Hb, Wb = H // 8, W // 8
    d = draw[:Hb*8, :Wb*8].reshape(Hb, 8, Wb, 8).transpose(0, 2, 1, 3).reshape(Hb, Wb, 64)
    p = prim[:Hb*8, :Wb*8].reshape(Hb, 8, Wb, 8).transpose(0, 2, 1, 3).reshape(Hb, Wb, 64)
    ds = np.sort(d, axis=2)
    distinct = 1 + (np.diff(ds, axis=2) != 0).sum(axis=2)
    has_bg = (ds[:, :, 0] == -1)
    ndraws = distinct - has_bg            # distinct real draws per block
    covered = (d >= 0).sum(axis=2)
    busy = ndraws > 0
    ps = np.sort(p, axis=2); tris = 1 + (np.diff(ps, axis=2) != 0).sum(axis=2) - has_bg
    print(f"{name}: {len(scene.draws)} draws; blocks with pixels {busy.sum()}, shading passes {ndraws.sum()} = {ndraws.sum()/busy.sum():.3f} per busy block; lanes busy per pass {covered.sum()/ndraws.sum():.1f} of 64; distinct triangles per busy block {tris[busy].mean():.1f}")
