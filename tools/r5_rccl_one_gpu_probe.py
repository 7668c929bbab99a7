import os, torch, torch.distributed as dist
r=int(os.environ["RANK"]); w=int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=r, world_size=w, device_id=torch.device("cuda",0))
    t=torch.ones(1024,device="cuda")*(r+1)
    dist.all_reduce(t); torch.cuda.synchronize()
    print("rank",r,"all_reduce ok", float(t[0]))
except Exception as e:
    print("rank",r,"FAILED:",str(e)[:300])
