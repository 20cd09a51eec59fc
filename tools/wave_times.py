#!/usr/bin/env python3
"""Start / end clock of every wave of k_render for one eighth of the headline frame: which pixels run
longest, when they were handed out, how long the launch waits for them.  Needs the debug build
`make -C raytracing_amd/csrc OUT=variants_wt.so EXTRA=-DRT_WAVE_TIMES` (the product library does not
carry the clock reads)."""
import os, sys, ctypes as C
os.environ["RTAMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracing_amd", "csrc", "variants_wt.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import api, host, parallel_render
api.init(0)
dev = torch.device("cuda", 0)
g = host.HostScene.builtin(2).upload()
w, h, spp = 1920, 1080, 64
cam = host.camera_default(w, h)
world = 8
lay = parallel_render.TileLayout(w, h, 16, 16, world)
out = torch.zeros(lay.buffer_floats(), dtype=torch.float32, device=dev)
nw = lay.per_rank * 256
buf = torch.zeros(nw * 3, dtype=torch.int64, device=dev)
L = api.lib()
L.rt_debug_wave_times.argtypes = [C.c_void_p]
assert L.rt_debug_wave_times(buf.data_ptr()) == 0
fn = parallel_render.device_render_fn(g, cam, w, h, spp, 1, 16, 16, None)
for rank in (0, 6):
    fn(out, rank, world); torch.cuda.synchronize()
    buf.zero_()
    fn(out, rank, world); torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 3)
    widx = np.nonzero(t[:, 1] > 0)[0]
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us at 100 MHz
    dur = en - st
    print(f"rank {rank}: waves {len(t)}, launch span {en.max():.0f} us, mean wave {dur.mean():.1f} us, max wave {dur.max():.0f} us")
    for i in np.argsort(-en)[:8]:
        print(f"   ends {en[i]:.0f} us  started {st[i]:.0f} us  ran {dur[i]:.0f} us  pixel ({t[i, 2] & 0xfff}, {(t[i, 2] >> 12) & 0xfff}) path nodes {(t[i, 2] >> 24) & 0xfffff} BVH nodes {((t[i, 2] >> 44) & 0xfffff) * 64} slot {widx[i] // 256}")
    print("   wave duration percentiles 50/90/99/99.9:", [round(float(np.percentile(dur, p)), 0) for p in (50, 90, 99, 99.9)])
    slow = dur > 500
    print("   slots of waves > 0.5 ms: min", (widx[slow] // 256).min(), "max", (widx[slow] // 256).max(), "distinct", len(set(widx[slow] // 256)))
    print("   waves > 1 ms:", int((dur > 1000).sum()), " > 2 ms:", int((dur > 2000).sum()), " > 3 ms:", int((dur > 3000).sum()))
