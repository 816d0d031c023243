"""How long does the host take to ENQUEUE one data-parallel step (4 graph segments + 2 collectives)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from distributeddeeplearning_amd.models.dcgan import GANConfig, GANTrainer, GraphedGANStep
from distributeddeeplearning_amd.parallel import ddp
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29688")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
for full in (False, True):
    t = GANTrainer(GANConfig(batch=1000), "cuda:0", ddp.FlatDataParallel(), 0)
    real = t.D.prepare_input(torch.rand(1000, 28, 28, 1))
    g = GraphedGANStep(t, real, capture_collectives=full)
    for _ in range(10):
        g.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"full={full}: host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step, wall {1e3 * (t2 - t0) / 50:.3f} ms/step", flush=True)
    # one segment at a time, synchronised: pure launch latency of each piece
    if not full:
        for i, gr in enumerate(g.graphs):
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(20):
                gr.replay()
            b = time.perf_counter()
            torch.cuda.synchronize()
            print(f"  segment {i}: host {1e6 * (b - a) / 20:.0f} us per replay", flush=True)
dist.destroy_process_group()
