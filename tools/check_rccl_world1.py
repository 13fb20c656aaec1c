"""Sanity check of the RCCL calls bench.py makes at N > 1 (init with device_id, gather, all_reduce MAX, barrier) on a one-GPU box:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 tools/check_rccl_world1.py"""
import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
t = torch.ones(4, 3, device=dev)
outs = [torch.empty_like(t)]
dist.gather(t, outs, dst=0)
x = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(x, op=dist.ReduceOp.MAX); dist.barrier()
print("nccl world 1 ok", outs[0].sum().item(), x.item())
dist.destroy_process_group()
