"""Dev probe: do gloo's collectives carry DEVICE tensors when two ranks share one GPU?  (They do on this image -- which is what
bench.py's SC_BENCH_SHARE_GPU rehearsal of the N > 1 path rests on; RCCL refuses two ranks on one device.)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/gpu_gloo_probe.py
"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
x = torch.full((4, 8), rank + 1, dtype=torch.int32, device="cuda:0")
out = torch.empty((world * 4, 8), dtype=torch.int32, device="cuda:0")
try:
    dist.all_gather_into_tensor(out, x)
    print(rank, "all_gather_into_tensor on cuda tensors via gloo:", out[:, 0].tolist(), flush=True)
except Exception as e:
    print(rank, "all_gather_into_tensor failed:", repr(e)[:300], flush=True)
    outs = [torch.empty_like(x) for _ in range(world)]
    try:
        dist.all_gather(outs, x); print(rank, "all_gather list ok", [int(o[0, 0]) for o in outs], flush=True)
    except Exception as e2:
        print(rank, "all_gather failed:", repr(e2)[:300], flush=True)
f = torch.tensor([1.5 * (rank + 1)], dtype=torch.float64, device="cuda:0")
g = torch.empty(world, dtype=torch.float64, device="cuda:0")
try:
    dist.all_gather_into_tensor(g, f); print(rank, "f64 gather", g.tolist(), flush=True)
except Exception as e:
    print(rank, "f64 gather failed", repr(e)[:200], flush=True)
dist.barrier()
dist.destroy_process_group()
