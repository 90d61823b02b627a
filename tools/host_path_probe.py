import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import libfriendship_amd
from libfriendship_amd import synth
import ctypes as C

V, P, T = 64, 4096, 4800
tree = synth.additive_tree(V, P)
r = libfriendship_amd.HipRenderer()
synth.install(r, tree)
idx = 0
def host_call():
    global idx
    t = synth.time_ramp(idx, idx + T)
    t0 = time.perf_counter()
    r.fill_buffer(V, idx, idx + T, [t])
    idx += T
    return (time.perf_counter() - t0) * 1e6
for _ in range(30): host_call()
print("host-buffer call: median %.1f us" % np.median([host_call() for _ in range(50)]))

# components: device call + sync; D2H into pageable vs pinned
d_out = torch.empty((V, T), dtype=torch.float32, device="cuda")
d_in = torch.zeros(T, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
def dev_call():
    global idx
    d_in.copy_(torch.from_numpy(synth.time_ramp(idx, idx + T)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.fill_buffer_device(d_out.data_ptr(), V, T, idx, d_in.data_ptr(), [0, T], stream)
    torch.cuda.synchronize()
    idx += T
    return (time.perf_counter() - t0) * 1e6
for _ in range(30): dev_call()
print("device call + sync: median %.1f us" % np.median([dev_call() for _ in range(50)]))
pageable = torch.empty((V, T), dtype=torch.float32)
pinned = torch.empty((V, T), dtype=torch.float32, pin_memory=True)
def d2h(dst):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(d_out, non_blocking=True); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
for dst, name in ((pageable, "pageable"), (pinned, "pinned")):
    for _ in range(10): d2h(dst)
    print("D2H 1.2 MB into %s: median %.1f us" % (name, np.median([d2h(dst) for _ in range(50)])))
a = np.empty((V, T), dtype=np.float32)
src = pinned.numpy()
ts = []
for _ in range(50):
    t0 = time.perf_counter(); np.copyto(a, src); ts.append((time.perf_counter() - t0) * 1e6)
print("CPU copy pinned -> pageable 1.2 MB: median %.1f us" % np.median(ts))
h = torch.from_numpy(synth.time_ramp(0, T))
hp = h.pin_memory()
def h2d(srct):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d_in.copy_(srct, non_blocking=True); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
for srct, name in ((h, "pageable"), (hp, "pinned")):
    for _ in range(10): h2d(srct)
    print("H2D 19 KB from %s: median %.1f us" % (name, np.median([h2d(srct) for _ in range(50)])))
