"""A co-tenant for tools/repro_flake*.py: another PROCESS keeping the same GPU busy with matrix products for the given number of seconds."""
import sys
import time

import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
x = torch.randn(n, n, device="cuda")
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(50):
        y = x @ x
    torch.cuda.synchronize()
