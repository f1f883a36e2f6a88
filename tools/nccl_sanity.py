"""Sanity check of the RCCL operations the multi-GPU path uses (process group, all_gather, fp64 / int64 all_reduce,
MAX) with a one-rank group: python tools/nccl_sanity.py on the GPU box."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t=torch.tensor([7],dtype=torch.int64,device=dev); out=[torch.zeros_like(t)]
dist.all_gather(out,t); print("all_gather", int(out[0].item()))
x=torch.ones(1000,dtype=torch.float64,device=dev); dist.all_reduce(x); print("all_reduce f64", float(x.sum()))
y=torch.ones(10,dtype=torch.int64,device=dev); dist.all_reduce(y); print("all_reduce i64", int(y.sum()))
z=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(z, op=dist.ReduceOp.MAX); print("max", float(z))
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group(); print("ok")
