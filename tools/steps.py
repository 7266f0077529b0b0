"""Print the per-step kernel times of the bench workload (diagnostic)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb
from jpeg_decoder_amd import synth
from jpeg_decoder_amd.api import torch_batch
W = H = 4096; nimg = 8
dev = torch.device("cuda:0"); st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
desc = jb.make_desc(W, H, 1, 1); g = jb.geometry_of(desc)
coef, q = synth.synth_blocks(W, H, 1, 1, 0)
base = torch.from_numpy(coef).to(dev)
coef_t = torch.stack([torch.roll(base, i * 7919 * 3, 0) for i in range(nimg)])
q_t = torch.from_numpy(jb.resolve_qtabs(desc, q)).to(dev)
rgb_t = torch.zeros((nimg, H, 3 * W), dtype=torch.uint8, device=dev)
b = torch_batch(desc, nimg, coef_t, q_t, rgb_t); ctx = jb.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record(st)
for i in range(n):
    ctx.blocks_to_rgb_device(b, st.cuda_stream); ev[i + 1].record(st)
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
print("per-step us:", " ".join("%.0f" % x for x in t[:60]), "...")
print("mean first 20: %.1f  mean last 100: %.1f  min %.1f  max %.1f" % (t[:20].mean(), t[-100:].mean(), t.min(), t.max()))
