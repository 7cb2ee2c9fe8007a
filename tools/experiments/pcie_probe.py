"""what the host link of the box does: pinned H2D alone, D2H alone, both at once on two streams (torch only).  python tools/experiments/pcie_probe.py"""
import time, torch
n = 256 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device="cuda"); d_dn = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn, reps=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
run(True, True, 2)
for name, up, dn in (("H2D alone", 1, 0), ("D2H alone", 0, 1), ("both at once", 1, 1)):
    t = run(up, dn)
    print("%-14s %6.1f ms per 256 MiB%s: %5.1f GB/s%s" % (name, t * 1e3, " each way" if up and dn else "", n / t / 1e9, " each way, %.1f total" % (2 * n / t / 1e9) if up and dn else ""))
# the same with a kernel that holds every CU busy on a third stream (the persistent walk does)
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
s3 = torch.cuda.Stream()
def busy():
    with torch.cuda.stream(s3):
        for _ in range(40): x.mul_(1.0001)
busy(); t = run(1, 1); print("both at once under elementwise kernels on a third stream: %.1f GB/s each way" % (n / t / 1e9))
