"""Dev tool: does the HIP runtime move a large pinned <-> device copy with an SDMA engine or with a blit kernel
(__amd_rocclr_copyBuffer), depending on the byte offset of the host side inside its pinned allocation?

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o p -- python3 tools/gpu_copy_probe.py d2h|h2d OFFSET

Three copies of 256 MiB on a side stream; count the SDMA records and the copyBuffer kernels in the traces."""
import sys

import torch

direction, off = sys.argv[1], int(sys.argv[2])
n = 256 << 20
host = torch.empty(n + 8192, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(3):
    with torch.cuda.stream(side):
        if direction == "d2h":
            host[off:off + n].copy_(dev, non_blocking=True)
        else:
            dev.copy_(host[off:off + n], non_blocking=True)
    side.synchronize()
print("done", direction, off)
