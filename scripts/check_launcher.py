"""Check the rendezvous bench.py uses for N > 1 (gloo broadcast of a 128-byte id) under torch.distributed.run."""
import os
import torch.distributed as dist
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
box = [bytes(range(128)) if rank == 0 else None]
dist.broadcast_object_list(box, src=0)
assert bytes(box[0]) == bytes(range(128))
dist.barrier()
print(f"rank {rank}/{world} ok, MASTER_PORT={os.environ.get('MASTER_PORT')}, agent store={os.environ.get('TORCHELASTIC_USE_AGENT_STORE')}")
