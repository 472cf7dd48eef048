#!/usr/bin/env python3
"""RCCL on a one-GPU box: a single-rank process group whose rank sends to itself (grouped isend + irecv on a side stream,
ordered against the compute stream with events, as pipeline._SideStreamLink / the ring hand-off do)."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist

def main():
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("backend", dist.get_backend(), "nccl version", torch.cuda.nccl.version(), flush=True)
    side = torch.cuda.Stream(device=dev)
    ok = True
    for it in range(3):
        src = torch.randn(1, 4, 14, 72, 128, device=dev, dtype=torch.float16) * (it + 1)
        dst = torch.zeros_like(src)
        ready = torch.cuda.Event(); ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)])
            for r in reqs: r.wait()
            done = torch.cuda.Event(); done.record()
        torch.cuda.current_stream().wait_event(done)
        src.record_stream(side); dst.record_stream(side)
        ok = ok and bool(torch.equal(src, dst))
    torch.cuda.synchronize()
    print("self-loop p2p ok" if ok else "MISMATCH", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)

if __name__ == "__main__":
    main()
