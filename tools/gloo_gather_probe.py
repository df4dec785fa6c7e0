import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import zero_chain_amd as zk
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B = 256
o = np.zeros(B * 192, dtype=np.uint8)
for k in range(6):
    if k % 2: time.sleep(0.3)
    t = time.perf_counter()
    zk.gather_proofs(o, B * world, dist=dist, device=torch.device("cpu"), dst=0)
    print(rank, k, round((time.perf_counter() - t) * 1e3, 2), flush=True)
dist.barrier(); dist.destroy_process_group()
