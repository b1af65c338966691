import torch, time
x = torch.empty((32, 4096, 257), dtype=torch.int32).pin_memory()
d = torch.empty_like(x, device="cuda")
for _ in range(3): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): d.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"H2D {x.numel()*4/1e6:.1f} MB pinned: {dt*1e3:.2f} ms = {x.numel()*4/dt/1e9:.1f} GB/s")
h = torch.empty_like(x).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): h.copy_(d, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"D2H: {dt*1e3:.2f} ms = {x.numel()*4/dt/1e9:.1f} GB/s")
