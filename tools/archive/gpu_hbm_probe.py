"""What this box's HBM delivers to plain streaming kernels (torch reductions / copies over 8 GB of fp64): the ceiling against
which the row-panel stream of k_chol_diag (3.3 TB/s during its K-loop, an ablation build of round 2) is to be read."""
import time
import torch
x = torch.ones(1 << 30, dtype=torch.float64, device="cuda")      # 8 GiB
y = torch.empty_like(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    b = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - b) / reps
gb = x.numel() * 8 / 1e9
print(f"sum (read 8.6 GB):        {gb / t(lambda: x.sum()) / 1e3:.2f} TB/s")
print(f"abs-max (read):           {gb / t(lambda: x.abs().max()) / 1e3:.2f} TB/s (reads + an 8.6 GB temporary)")
print(f"copy (read + write):      {2 * gb / t(lambda: y.copy_(x)) / 1e3:.2f} TB/s")
print(f"fill (write):             {gb / t(lambda: y.fill_(2.0)) / 1e3:.2f} TB/s")
print(f"dot (read 2 x 8.6 GB):    {2 * gb / t(lambda: torch.dot(x, y)) / 1e3:.2f} TB/s")
