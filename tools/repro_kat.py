import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import jpeg_decoder_amd as jb
from conftest import load_kat
kat = load_kat()
names = sys.argv[1:] or [n for n in kat if "2x2" in n]
with jb.Context(0, 64 << 20, 64 << 20, 2) as ctx:
    for n in names:
        desc, coef, q, rgb = kat[n]
        d = jb.make_desc(desc.width, desc.height, desc.hs, desc.vs, list(desc.qtab_id))
        bad = 0; where = set()
        for rep in range(50):
            got = ctx.blocks_to_rgb(d, coef, q)
            if not np.array_equal(got, rgb):
                bad += 1
                ys, xs, cs = np.nonzero(got != rgb)
                where |= set(zip(ys.tolist(), xs.tolist(), cs.tolist()))
        print(n, f"{desc.width}x{desc.height}", "bad runs:", bad, "/50", sorted(where)[:12])
