import os, time, numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
est = np.random.rand(5000).astype(np.float32)
def T(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def full():
    mine = torch.from_numpy(est).cuda()
    allv = [torch.empty_like(mine) for _ in range(1)]
    dist.all_gather(allv, mine)
    return torch.cat(allv).cpu().numpy()
print("full us", T(full))
mine = torch.from_numpy(est).cuda()
print("h2d us", T(lambda: torch.from_numpy(est).cuda()))
allv = [torch.empty_like(mine)]
print("all_gather(list) us", T(lambda: dist.all_gather(allv, mine)))
out = torch.empty(5000, dtype=torch.float32, device="cuda")
print("all_gather_into_tensor us", T(lambda: dist.all_gather_into_tensor(out, mine)))
print("d2h us", T(lambda: out.cpu()))
pin_in = torch.empty(5000, dtype=torch.float32).pin_memory(); pin_out = torch.empty(5000, dtype=torch.float32).pin_memory()
def fast():
    pin_in.numpy()[:] = est
    mine.copy_(pin_in, non_blocking=True)
    dist.all_gather_into_tensor(out, mine)
    pin_out.copy_(out, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return pin_out.numpy()
print("fast us", T(fast))
dist.destroy_process_group()
