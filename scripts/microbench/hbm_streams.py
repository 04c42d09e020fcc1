"""What the HBM of this box sustains for pure streams, as reference points for the rooflines in DESIGN.md §4:
fill (write only), copy (read + write) and reduce (read only) over buffers far beyond the 256 MiB Infinity Cache.
k_decode_terms writes 1024 B and reads ~150 B per block: it should be read against the WRITE-dominated figures.
usage: python scripts/microbench/hbm_streams.py [GiB]   (torch only provides the device buffers and the stock kernels)"""
import sys
import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30)) // 4
a = torch.empty(n, dtype=torch.int32, device="cuda")
b = torch.empty(n, dtype=torch.int32, device="cuda")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


nbytes = n * 4
t = timed(lambda: a.fill_(7))
print("fill   (write only)   %7.1f GB/s" % (nbytes / t / 1e9))
t = timed(lambda: b.copy_(a))
print("copy   (read + write) %7.1f GB/s  (sum of both directions)" % (2 * nbytes / t / 1e9))
t = timed(lambda: a.sum())
print("reduce (read only)    %7.1f GB/s" % (nbytes / t / 1e9))
# 7 : 1 write : read mix, the shape of the materialising decode (one int32 in -> eight int32 out per lane)
src = a[: n // 8]
t = timed(lambda: torch.repeat_interleave(src, 8, output_size=n // 8 * 8))
print("expand (1 read : 8 write) %7.1f GB/s" % ((n // 8 * 4 + n // 8 * 8 * 4) / t / 1e9))
