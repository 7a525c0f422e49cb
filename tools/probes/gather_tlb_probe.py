"""Is a process's device memory mapped with small pages? Random 128-byte row gathers over a 128 MB table (torch.index_select) and a
streaming copy, a few times; run in several processes: a bimodal gather rate with a steady copy rate points at the mapping."""
import time, torch
rows = 1 << 20
table = torch.randn(rows, 32, device="cuda")
idx = torch.randint(0, rows, (1 << 23,), device="cuda")
out = torch.empty(1 << 23, 32, device="cuda")
a = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for r in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(10): torch.index_select(table, 0, idx, out=out)
    torch.cuda.synchronize(); g = time.perf_counter() - t
    t = time.perf_counter()
    for i in range(20): b.copy_(a)
    torch.cuda.synchronize(); c = time.perf_counter() - t
print("gather %.0f GB/s   copy %.0f GB/s" % (10 * (1 << 23) * 128 * 2 / g / 1e9, 20 * 2 * (1 << 28) / c / 1e9))
