#!/usr/bin/env python
"""Does RCCL accept two ranks on ONE device?  (It decides how the two-rank HIP
test can exchange partials on a one-GPU box.)  Run under `timeout`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, world, port):
    import torch
    import torch.distributed as dist

    from cotengra_amd import runtime

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box = [runtime.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        c = runtime.Comm(box[0], rank, world, 0)
        print(rank, "comm ok", flush=True)
        c.close()
    except Exception as e:
        print(rank, "comm refused:", type(e).__name__, e, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    mp.spawn(main, args=(2, 29617), nprocs=2)
