"""Dependent-launch cost of a chain of small kernels on one stream: plain launches against a captured hipGraph replay (is the trunk's forward, ~100
dependent launches of 5-40 us, worth capturing?)."""
import torch, time
dev = "cuda:0"
x = torch.zeros(1 << 16, device=dev)
def chain(n):
    for _ in range(n):
        x.add_(1.0)
N = 200
for _ in range(3): chain(N)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); chain(N); e1.record(); torch.cuda.synchronize()
print("stream: %.2f us per dependent launch" % (1e3 * e0.elapsed_time(e1) / N))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(3)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain(N)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("graph:  %.2f us per dependent launch" % (1e3 * e0.elapsed_time(e1) / N))
# bigger kernels (20 us each): does the gap survive?
y = torch.zeros(32 << 20, device=dev)
def chain2(n):
    for _ in range(n):
        y.add_(1.0)
for _ in range(2): chain2(50)
torch.cuda.synchronize()
e0.record(); chain2(100); e1.record(); torch.cuda.synchronize()
t1 = 1e3 * e0.elapsed_time(e1) / 100
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s):
        chain2(100)
torch.cuda.synchronize()
g2.replay(); torch.cuda.synchronize()
e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
print("128 MB add_: stream %.2f us, graph %.2f us per launch" % (t1, 1e3 * e0.elapsed_time(e1) / 100))
# the same chain behind a long-running kernel, so that the host is far ahead when the GPU reaches it (the training step's situation): GPU-side gap only
big = torch.zeros(512 << 20, device=dev)
for rep in range(2):
    for _ in range(6): big.add_(1.0)           # ~6 x 1 ms
    e0.record(); chain(N); e1.record(); torch.cuda.synchronize()
    print("stream, queue pre-filled: %.2f us per dependent launch" % (1e3 * e0.elapsed_time(e1) / N))
for rep in range(2):
    for _ in range(6): big.add_(1.0)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("graph, queue pre-filled:  %.2f us per dependent launch" % (1e3 * e0.elapsed_time(e1) / N))
