import numpy as np, sys
sys.path.insert(0, '/root/repo')
from superviseddescent_amd import Context
rows, factor_tiles, rhs_tiles = int(sys.argv[1]), 5, 1
rng = np.random.default_rng(1)
wf, w = 128 * factor_tiles, 128 * (factor_tiles + rhs_tiles)
P = rng.standard_normal((rows, w)).astype(np.float32)
P[:, wf:] *= 37.0
C = (rng.standard_normal((w, w)) * 10.0).astype(np.float32)
bound = float(np.abs(P[:, :wf]).max()) ** 2 * 1.01
ctx = Context(0)
outs = [ctx.debug_update_f16(P, C, wf, bound) for _ in range(4)]
want = C.astype(np.float64) - P.astype(np.float64).T @ P.astype(np.float64)
ti, tj = np.arange(w)[:, None] // 128, np.arange(w)[None, :] // 128
written = (ti <= tj) & (ti < factor_tiles)
for k, o in enumerate(outs):
    d = o != outs[0]
    bad = np.argwhere(d)
    err = np.abs(o - want)
    print("run", k, "differs from run 0 in", int(d.sum()), "entries; tiles", sorted(set((int(a) // 128, int(b) // 128) for a, b in bad[:2000]))[:12],
          "| max err in written", float(err[written].max()), "| unwritten changed", int((o[~written] != C[~written]).sum()), "nan", int(np.isnan(o).sum()))
    if d.any():
        a, b = bad[0]
        print("   first diff at", a, b, o[a, b], outs[0][a, b], want[a, b], "rows in tile", sorted(set(int(x) % 128 for x, _ in bad[:500]))[:20], "cols", sorted(set(int(y) % 128 for _, y in bad[:500]))[:20])
o = outs[0]
ch = (o != C) & ~written
bad = np.argwhere(ch)
tiles = {}
for a, b in bad:
    tiles.setdefault((int(a) // 128, int(b) // 128), set()).add(int(a) % 128)
for t, rws in sorted(tiles.items()):
    rws = sorted(rws)
    print("changed outside the mask: tile", t, "rows", rws[0], "..", rws[-1], "count", len(rws), "| matches C - P^T P:", bool(np.allclose(o[t[0]*128+rws[0], t[1]*128:(t[1]+1)*128], want[t[0]*128+rws[0], t[1]*128:(t[1]+1)*128], rtol=1e-3, atol=1e-2)))
