#!/usr/bin/env python3
"""Which HIP streams share a hardware queue?  (ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default -- and two
streams on one queue serialise: a kernel on one waits for the kernels queued earlier on the other.)

    python tools/queue_probe.py [n_streams]

For every pair (X, Y) of {null stream, n pool streams}: a ~300 us spin on X, then a tiny kernel on Y; Y's kernel finishing only when X's
spin does means the two share a queue.  Prints the conflict classes."""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(n)]
names = ["null"] + ["s%d" % i for i in range(n)]
x = torch.zeros(1024, device=dev)
SPIN = 30_000_000        # torch.cuda._sleep cycles; calibrated below
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(SPIN); e1.record(); torch.cuda.synchronize()
spin_ms = e0.elapsed_time(e1)
SPIN = int(SPIN * 0.3 / spin_ms)
e0.record(); torch.cuda._sleep(SPIN); e1.record(); torch.cuda.synchronize()
spin_ms = e0.elapsed_time(e1)
print("spin = %.3f ms" % spin_ms)


def conflict(a, b):
    sa, sb = streams[a], streams[b]
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True); done = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        start.record(sa)
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(sb):
        x.add_(1.0)
        done.record(sb)
    torch.cuda.synchronize()
    return start.elapsed_time(done) > 0.6 * spin_ms


N = len(streams)
conf = [[False] * N for _ in range(N)]
for i in range(N):
    for j in range(N):
        if i != j:
            conf[i][j] = conflict(i, j)
cls = []
for i in range(1, N):                  # (the null stream apart: work on any stream may also wait for it by the legacy-stream rule)
    for c in cls:
        if conf[c[0]][i] and conf[i][c[0]]:
            c.append(i)
            break
    else:
        cls.append([i])
for c in cls:
    print("queue class:", " ".join(names[i] for i in c))
print("spin on null delays:", " ".join(names[j] for j in range(1, N) if conf[0][j]))
print("spin on these delays null:", " ".join(names[i] for i in range(1, N) if conf[i][0]))
for i in range(N):
    print("%5s " % names[i] + "".join("X" if conf[i][j] else "." for j in range(N)))
asym = [(names[i], names[j]) for i in range(N) for j in range(N) if conf[i][j] != conf[j][i]]
print("asymmetric pairs:", asym[:10])
print("cuda_stream ids:", {names[i]: hex(streams[i].cuda_stream) for i in range(N)})
